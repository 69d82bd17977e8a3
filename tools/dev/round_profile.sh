#!/bin/bash
# End-of-round measurement set (run ON the GPU box): GPU test log, bench line, rocprofv3
# kernel stats of the same command, HBM PMC traffic, voxel2obj bench / stats / PMC, the
# other configs.   usage: bash tools/dev/round_profile.sh r02 [tests|bench|pmc|unet|other ...]
set -o pipefail
tag=${1:-rXX}
shift
what=${*:-tests bench pmc unet other}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for w in $what; do
case $w in
tests)
  python -m pytest tests -m gpu -q > $out/${tag}_tests_gpu.log 2>&1; tail -2 $out/${tag}_tests_gpu.log ;;
bench)
  # the headline: BASELINE.json's 520^3 volume (bench.py's default), then configs[1]'s 1024^3
  python bench.py > $out/${tag}_bench520_f16s.json 2> $out/bench_f16s.err
  cut -c1-200 $out/${tag}_bench520_f16s.json
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --no-legs --no-cpu-baseline --steps 20 > $out/trace_bench.json 2> $out/trace.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_bench520_f16s_kernel_stats.csv \;
  rm -rf $out/trace
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o bench -- python3 bench.py --size 1024 --no-legs --no-cpu-baseline > $out/trace_bench1024.json 2> $out/trace.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_bench1024_f16s_kernel_stats.csv \;
  rm -rf $out/trace
  python tools/bench_v2o.py --reps 10 --out $out/${tag}_v2o582_bench.json > $out/v2o.log 2>&1; tail -1 $out/v2o.log
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o v2o -- python3 tools/bench_v2o.py --reps 10 > $out/trace_v2o.json 2> $out/trace_v2o.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_v2o582_kernel_stats.csv \;
  rm -rf $out/trace
  echo bench done ;;
pmc)
  python tools/profile_pmc.py --size 520 --precision f16s --steps 4 --out $out/pmc520 > $out/pmc520.log 2>&1 && \
    cp $out/pmc520/summary.json $out/${tag}_pmc_hbm_520_f16s.json && cp $out/pmc520/summary.md $out/${tag}_pmc_hbm_520_f16s.md
  rm -rf $out/pmc520
  python tools/profile_pmc.py --size 1024 --precision f16s --out $out/pmc > $out/pmc.log 2>&1 && \
    cp $out/pmc/summary.json $out/${tag}_pmc_hbm_1024_f16s.json && cp $out/pmc/summary.md $out/${tag}_pmc_hbm_1024_f16s.md
  python tools/profile_pmc.py --size 1024 --precision f16 --out $out/pmc16 > $out/pmc16.log 2>&1 && \
    cp $out/pmc16/summary.json $out/${tag}_pmc_hbm_1024_f16.json && cp $out/pmc16/summary.md $out/${tag}_pmc_hbm_1024_f16.md
  rm -rf $out/pmc16
  python tools/profile_pmc.py --target v2o --size 582 --out $out/pmc_v2o > $out/pmc_v2o.log 2>&1 && \
    cp $out/pmc_v2o/summary.json $out/${tag}_v2o582_pmc.json
  rm -rf $out/pmc $out/pmc_v2o
  echo pmc done ;;
unet)
  # unet_like2 (27 tiles of 100^3): counters and kernel stats of the four precisions
  python tools/profile_pmc.py --target unet --size 264 --out $out/pmc_unet > $out/pmc_unet.log 2>&1 && \
    cp $out/pmc_unet/summary.json $out/${tag}_pmc_unet264.json && cp $out/pmc_unet/summary.md $out/${tag}_pmc_unet264.md
  rm -rf $out/pmc_unet
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o unet -- python3 tools/bench_configs.py --what unet > $out/trace_unet.json 2> $out/trace_unet.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_unet264_kernel_stats.csv \;
  rm -rf $out/trace
  echo unet done ;;
other)
  python tools/bench_configs.py --out $out/${tag}_other_configs.json > $out/other.log 2>&1; tail -2 $out/other.log
  python tools/bench_configs.py --what roi --out $out/${tag}_roi1536.json > $out/roi.log 2>&1; tail -1 $out/roi.log
  python tools/bench_configs.py --what roi --roi-precision f16 --out $out/${tag}_roi1536_f16.json > $out/roi16.log 2>&1; tail -1 $out/roi16.log
  # every kernel of every lane of the pipeline (HIP events on six concurrent streams measure the waits for one
  # another, not execution): rocprofv3's kernel trace of the same run, oracle check skipped
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o roi -- python3 tools/bench_configs.py --what roi --skip-oracle > $out/trace_roi.json 2> $out/trace_roi.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_roi1536_kernel_stats.csv \;
  python tools/dev/trace_union.py $out/trace --last-span-ms 430 > $out/${tag}_roi1536_trace_union.txt 2>&1
  rm -rf $out/trace
  tail -1 $out/trace_roi.json
  # the same ROI kept RESIDENT in HBM (inputs in place when the clock starts), and its trace by kernel class
  python tools/bench_configs.py --what roi --roi-source resident --out $out/${tag}_roi1536_resident.json > $out/roi_res.log 2>&1; tail -1 $out/roi_res.log
  rocprofv3 --kernel-trace --output-format csv -d $out/trace -o roi -- python3 tools/bench_configs.py --what roi --roi-source resident --skip-oracle > $out/trace_roi_res.json 2> $out/trace_roi_res.err
  python tools/dev/trace_union.py $out/trace --last-span-ms 430 > $out/${tag}_roi1536_resident_trace_union.txt 2>&1
  rm -rf $out/trace
  # the four factories of the graph executor: 'auto' / f16 / f32, and the kernel stats of that run
  python tools/bench_configs.py --what graphs 2>&1 | grep -v "^#\|^[0-9]* \|total params\|amdgpu.ids" > $out/${tag}_graph_executor_four_factories.txt
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o graphs -- python3 tools/bench_configs.py --what graphs > $out/trace_graphs.json 2> $out/trace_graphs.err
  find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/${tag}_graph_executor_kernel_stats.csv \;
  rm -rf $out/trace
  # the public API host to host (pool, pipelined copies): where its time goes
  python tools/dev/host_api_rate.py 520 2>&1 | grep -v "^#\|^[0-9]* \|total params\|amdgpu.ids" > $out/${tag}_host_api_520.txt
  tail -3 $out/${tag}_host_api_520.txt ;;
esac
done
