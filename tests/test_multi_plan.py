"""rayrs_render_multi's reduce on a node of N distinct GPUs (rayrs_amd/csrc/multi_device.cpp), rehearsed on the CPU.

No box of this pool has more than one GPU, so the device grouping, the communicator cache and the grouped RCCL calls have
only ever run with one device (tests/test_gpu_render.py forces the one-device communicator).  rayrs_lab_multi_rehearse runs the
same planning and the same reduce routine against a RECORDING table in place of librccl and the HIP runtime: what would be
called, in which order, for 2, 4 and 8 distinct devices, for ranks that share devices, over several frames, and when a
collective fails.  (What a real communicator does with those calls stays unmeasured: README.md says so.)"""
import ctypes as C

import pytest

from rayrs_amd import _ffi


def rehearse(devices, rounds=1, fail_at=-1):
    L = _ffi.lib()
    arr = (C.c_int * len(devices))(*devices)
    buf = C.create_string_buffer(1 << 16)
    st = L.rayrs_lab_multi_rehearse(arr, len(devices), rounds, fail_at, buf, len(buf))
    return st, [x for x in buf.value.decode().split(";") if x]


@pytest.mark.parametrize("n", [2, 4, 8])
def test_distinct_devices_one_communicator_set_one_grouped_reduce_per_frame(n):
    devs = list(range(n))
    st, log = rehearse(devs, rounds=3)
    assert st == 0
    dev_list = ",".join(map(str, devs))
    # the communicators of a device list are created once, at its first use, for exactly that list in rank order
    assert [x for x in log if x.startswith("init")] == [f"init[{dev_list}]"]
    assert log[-1] == "communicator_sets=1"
    frames = "|".join(log).split("plan ")[1:]
    assert len(frames) == 3
    for f in frames:
        calls = f.split("|")
        assert calls[0] == "devs[" + ",".join(f"{d}:{d}" for d in devs) + "] local[]"    # every rank leads its own device
        calls = [c for c in calls[1:] if c and not c.startswith("init") and not c.startswith("communicator_sets")]
        # group_start, then per device set_device + one in-place sum-reduce to root 0 on that device's communicator, group_end,
        # then every device's stream is waited for
        expect = ["group_start"]
        for k, d in enumerate(devs):
            expect += [f"set_device({d})", f"reduce(comm={k + 1},root=0,count=12,in_place=1,sum=1)"]
        expect += ["group_end"]
        for d in devs:
            expect += [f"set_device({d})", "sync"]
        expect += ["status=0"]
        assert calls == expect


def test_ranks_that_share_devices_are_summed_there_and_the_leaders_reduce():
    st, log = rehearse([2, 2, 5, 2, 7, 5])
    assert st == 0
    assert log[0] == "plan devs[2:0,5:2,7:4] local[0<-1,0<-3,1<-5]"   # device:leading rank; device index <- rank summed on it
    assert "init[2,5,7]" in log and sum(x.startswith("reduce(") for x in log) == 3
    st, log = rehearse([3, 3, 3, 3], rounds=2)
    assert st == 0 and log.count("one device: no collective") == 2 and not any(x.startswith("init") for x in log)
    assert log[-1] == "communicator_sets=0"


def test_another_device_list_gets_its_own_communicators_and_a_failed_collective_is_reported():
    # the second ncclReduce of the first frame fails: the group is closed, RAYRS_RCCL_ERROR comes back, the next frame
    # uses the cached communicators again
    st, log = rehearse([0, 1, 2, 3], rounds=2, fail_at=1)
    assert st == -7
    first = log[:log.index("status=-7") + 1]
    assert first.count("group_start") == 1 and first.count("group_end") == 1 and sum(x.startswith("reduce(") for x in first) == 2
    assert "sync" not in first
    second = log[log.index("status=-7") + 1:]
    assert "status=0" in second and not any(x.startswith("init") for x in second)
    assert log[-1] == "communicator_sets=1"
