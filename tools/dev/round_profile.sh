#!/bin/bash
# End-of-round measurement set (run ON the GPU box): bench line, rocprofv3 kernel stats
# of the same command, HBM PMC traffic, the other configs, the GPU test log.
# usage: bash tools/dev/round_profile.sh r01f
set -o pipefail
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q > $out/${tag}_tests_gpu.log 2>&1; tail -2 $out/${tag}_tests_gpu.log
python tools/profile_pmc.py --size 1024 --precision f16 --out $out/pmc > $out/pmc.log 2>&1 && \
  cp $out/pmc/summary.json profiles/${tag}_pmc_hbm_1024_f16.json && cp $out/pmc/summary.md profiles/${tag}_pmc_hbm_1024_f16.md && \
  cp profiles/${tag}_pmc_hbm_1024_f16.* $out/
echo pmc done
for prec in f16 bf16 f32; do
  python bench.py --precision $prec > $out/${tag}_bench1024_$prec.json 2> $out/bench_$prec.err
  cut -c1-200 $out/${tag}_bench1024_$prec.json
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python bench.py --precision f16 > $out/trace_bench.json 2> $out/trace.err
find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_bench1024_f16_kernel_stats.csv \;
echo trace done
python tools/bench_configs.py --out $out/${tag}_other_configs.json > $out/other.log 2>&1; tail -2 $out/other.log
python tools/bench_configs.py --what unet --unet-size 756 > $out/unet756.log 2>&1; head -c 600 $out/unet756.log
