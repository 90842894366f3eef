set -e
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2h_tests.log 2>&1; tail -3 gpurun_out/r2h_tests.log
timeout -k 10 300 bash tools/profile.sh r2h_c2 c2 2 > gpurun_out/r2h_prof_c2.log 2>&1; tail -1 gpurun_out/r2h_prof_c2.log
timeout -k 10 300 bash tools/profile.sh r2h_c3 c3 2 > gpurun_out/r2h_prof_c3.log 2>&1; tail -1 gpurun_out/r2h_prof_c3.log
timeout -k 10 300 bash tools/profile.sh r2h_c5tile c5tile 2 > gpurun_out/r2h_prof_c5tile.log 2>&1; tail -1 gpurun_out/r2h_prof_c5tile.log
timeout -k 10 300 bash tools/profile.sh r2h_c5frame c5frame 1 > gpurun_out/r2h_prof_c5frame.log 2>&1; tail -1 gpurun_out/r2h_prof_c5frame.log
