#!/bin/bash
# Developer tool (GPU box, via gpurun): rocprofv3 evidence for profiles/rNN -- kernel-trace stats of both bench workloads
# (single launch lane, so a launch's wall time is its own), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and
# MFMA-busy counters per kernel.  usage: tools/collect_profiles.sh <out_dir under gpurun_out>
set -o pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIPAC_LANES=1
B="python3 bench.py --steps 4 --warmup 1 --no_cpu_baseline --no_wsi --no_simclr"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_bench" -- $B > "$OUT/bench_lanes1_under_rocprof.json" 2> "$OUT/kt_bench.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_wsi" -- python3 bench.py --workload wsi --steps 2 --warmup 1 --no_cpu_baseline > "$OUT/wsi50k_under_rocprof.json" 2> "$OUT/kt_wsi.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_simclr" -- python3 bench.py --workload simclr --steps 1 --warmup 1 --simclr_views 256 > "$OUT/simclr256_under_rocprof.json" 2> "$OUT/kt_simclr.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_simclr_amp" -- python3 bench.py --workload simclr --steps 1 --warmup 1 --simclr_views 1024 --train_precision fp16 > "$OUT/simclr1024_amp_under_rocprof.json" 2> "$OUT/kt_simclr_amp.err" || exit 1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_tiff" -- python3 tools/tiffbench.py 40000 jpeg 16 > "$OUT/tiff40k_under_rocprof.txt" 2> "$OUT/kt_tiff.err" || exit 1
P="python3 bench.py --steps 2 --warmup 1 --no_cpu_baseline --no_wsi --no_simclr"
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS"; do
  n=$(echo $c | cut -d" " -f1)
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$n" -- $P > /dev/null 2> "$OUT/pmc_$n.err" || exit 1
done
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmcwsi_$c" -- python3 bench.py --workload wsi --steps 1 --warmup 1 --no_cpu_baseline > /dev/null 2> "$OUT/pmcwsi_$c.err" || exit 1
done
python3 tools/summarize_profiles.py "$OUT"
