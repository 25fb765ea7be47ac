# headline frame with sample chunks of 4 (the frame rule), 8 and 16, same box, order 4 8 16 16 8 4
python - <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "scripts"))
import perf_probe
for c in (4, 8, 16, 16, 8, 4):
    perf_probe.run(5, 2048, 2048, 1024, chunk=c)
PY
