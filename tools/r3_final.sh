#!/bin/bash
# round-3 final evidence (GPU box, repo root): full GPU suite, then the profile / bench collection
set -o pipefail
mkdir -p gpurun_out/r3final
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r3final/pytest.log 2>&1 || { tail -40 gpurun_out/r3final/pytest.log; exit 1; }
tail -3 gpurun_out/r3final/pytest.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r3final/smoke.log 2>&1 || { tail -20 gpurun_out/r3final/smoke.log; exit 1; }
tail -1 gpurun_out/r3final/smoke.log
bash tools/r3_profiles.sh
