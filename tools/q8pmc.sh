set -o pipefail
OUT=gpurun_out/q8pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIPAC_LANES=1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d "$OUT/pmc_$c" -- python3 tools/opbench.py fp16q8 2 > /dev/null 2> "$OUT/pmc_$c.err" || exit 1
done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_bench" -- python3 tools/opbench.py fp16q8 4 > $OUT/op.log 2> "$OUT/kt.err" || exit 1
python3 tools/summarize_profiles.py "$OUT" > /dev/null 2>&1
ls $OUT
