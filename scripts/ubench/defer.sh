# A/B of the traversal's leaf queue (a lane sets reached leaves aside and walks on; leaf phases run when
# most lanes have one waiting) on the headline frame, after checking that it renders the same bits.
python - <<'PY' || exit 1
import numpy as np, rayrs_amd
from rayrs_amd import scenes, procedural
cam_args, objs, heur, _, mb = scenes.config(5)
cam_args = scenes.camera_for_resolution(cam_args, 256, 256)
hdri = procedural.make_hdri(1024, 512)
scene = rayrs_amd.Scene(objs, 1e-6, 1e6, heur, hdri, device=0)
cam = rayrs_amd.Camera(*cam_args)
a, sa = rayrs_amd.render(scene, cam, 16, mb, sample_chunk=4)
for lm in (256 + 24, 1024 + 48, 1024 + 60, 512 + 40):
    scene.set_tuning(leaf_min=lm, stack_lds=8)
    b, sb = rayrs_amd.render(scene, cam, 16, mb, sample_chunk=4)
    assert np.array_equal(a, b) and sa['rays'] == sb['rays'], lm
print("leaf-queue frames identical")
PY
for T in "leaf_min=24" "stack_lds=8,leaf_min=1064" "stack_lds=8,leaf_min=1072" "stack_lds=8,leaf_min=1080" "stack_lds=8,leaf_min=1084" "stack_lds=8,leaf_min=560" "stack_lds=8,leaf_min=568" "stack_lds=8,leaf_min=816" "stack_lds=8,leaf_min=1072,refill_min=58"; do
  echo "== tuning: $T"
  PROBE_TUNING="$T" python scripts/perf_probe.py full5 2>&1 | tail -1
done
PROBE_TUNING="stack_lds=8,leaf_min=1072" python scripts/perf_probe.py u5 2>&1 | tail -3
