set -x
python scripts/exp_handover.py > gpurun_out/exp_handover.log 2>&1; tail -9 gpurun_out/exp_handover.log
python -m pytest tests/test_gpu_handover.py tests/test_gpu_tail_fused.py tests/test_gpu_rehearsal.py -q -x 2>&1 | tail -4
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 700 gpurun_out/bench_default.json
bash scripts/profile_round.sh r06 > gpurun_out/prof_r06.log 2>&1; tail -c 400 gpurun_out/prof_r06/bench_under_trace.json
python scripts/fullsize_reference_parity.py > gpurun_out/parity.log 2>&1; tail -3 gpurun_out/parity.log | cut -c1-300
python scripts/exp_train3.py > gpurun_out/train3_switches.json 2>/dev/null
