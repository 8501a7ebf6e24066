#!/bin/bash
# One GPU-box visit: debug + parity tests + sweep + bench + rocprofv3 kernel traces + PMC passes.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out
mkdir -p $O

python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest_exit=$?"
tail -4 $O/pytest_gpu.log
python scripts/sweep.py > $O/sweep.log 2>&1; echo "sweep_exit=$?"
python bench.py > $O/bench_c2.json 2> $O/bench_c2.err; echo "bench_c2_exit=$?"
python bench.py --workload c5 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err; echo "bench_c5_exit=$?"
for w in c2 c5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-extras > $O/prof_$w.log 2>&1; echo "prof_${w}_exit=$?"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_fetch_$w.log 2>&1; echo "pmc_fetch_${w}_exit=$?"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$w -- python3 bench.py --workload $w --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $O/pmc_write_$w.log 2>&1; echo "pmc_write_${w}_exit=$?"
done

rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pre -- python3 scripts/bench_preprocess.py > $O/prof_pre.log 2>&1; echo "prof_pre_exit=$?"
python scripts/bench_preprocess.py > $O/preprocess_times.txt 2>&1
echo done
