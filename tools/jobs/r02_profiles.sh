#!/bin/bash
# round-2 evidence for profiles/: kernel stats of the default bench, HBM + SQ/TA counters of the tiled gather
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r02 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/r02_bench_under_rocprof.json 2> $R/gpurun_out/r02_bench_under_rocprof.err
f=$(find $R/gpurun_out/prof_r02 -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/r02_bench_cfg2_kernel_stats.csv
t=$(find $R/gpurun_out/prof_r02 -name "*kernel_trace.csv" | head -1)
python3 - "$t" > $R/gpurun_out/r02_kernels_from_trace.json <<'PY'
import csv, json, sys
csv.field_size_limit(1 << 30)
rows = list(csv.DictReader(open(sys.argv[1])))
out = {}
for key in ("dfa3d_fwd_tile_kernel", "conv3d_halo_bf16x3_kernel<4, 4, 16>", "topk_select_kernel", "bin_place_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]]
    if d:
        d.sort()
        out[key] = dict(launches=len(d), avg_us=round(sum(d) / len(d), 2), median_us=round(d[len(d) // 2], 2), max_us=round(d[-1], 2))
print(json.dumps(out, indent=1))
PY
rm -rf $R/gpurun_out/prof_r02
for C in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/pmc_tile_$C -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/pmc_tile_$C.log 2>&1
echo rc $?
done
N=1; timeout 300 rocprofv3 --pmc TA_TA_BUSY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_tile_sq1 -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/pmc_tile_sq1.log 2>&1; echo rc $?
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_tile_sq2 -- python3 $R/bench.py --graph tail --streams 1 --steps 6 --warmup 2 --no-cpu-baseline --no-strict-fp32 --sustain 0 > $R/gpurun_out/pmc_tile_sq2.log 2>&1; echo rc $?
cd $R
for n in FETCH_SIZE WRITE_SIZE sq1 sq2; do python tools/pmc_summary.py gpurun_out/pmc_tile_$n "dfa3d_fwd_tile_kernel" 0 > gpurun_out/r02_pmc_tile_$n.json; rm -rf gpurun_out/pmc_tile_$n; done
cat gpurun_out/r02_pmc_tile_FETCH_SIZE.json gpurun_out/r02_pmc_tile_WRITE_SIZE.json gpurun_out/r02_kernels_from_trace.json
