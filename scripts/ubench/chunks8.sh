# one rank's eighth of the headline frame with sample chunks of 4, 8, 16 (same box, 4 8 16 16 8 4)
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
import perf_probe
for c in (4, 8, 16, 16, 8, 4):
    perf_probe.run(5, 2048, 2048, 1024, chunk=c, ranks=8)
for c in (4, 8, 8, 4):
    perf_probe.run(3, 1024, 1024, 512, chunk=c)
for c in (4, 8, 8, 4):
    perf_probe.run(2, 1024, 1024, 256, chunk=c)
PY
