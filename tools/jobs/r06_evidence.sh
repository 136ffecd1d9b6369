#!/bin/bash
# round-6 job 1 (one box): today's baseline of the unchanged product kernels + the evidence the round-5 review asked for:
#   1a  SQ counters + timing-knob decomposition of the Winograd-form halo convolution (90-GF layer)
#   1c  per-layer A/B of the form on the 20x20x8 layers WITH FOUR SCENES IN FLIGHT
#   2   SQ counters of the Cm = 16 tiled gather (config-4 finest level)
#   3   config 2 at the reference's 100 test views: parity test + bench line
#   7   power / clocks beside the headline run and the matrix pipe's power-limited rate on this tree
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd $R
filt() { grep -v "amdgpu.ids\|warn\|Warning"; }
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_base_cfg2_driver_cmd.json 2>/dev/null; echo base rc $?
# ---- 1a: decomposition (alternated rounds, one process) ----
timeout 900 python tools/wz_skip.py 2>&1 | filt > gpurun_out/r06_wz_skip.txt; echo wz_skip rc $?
timeout 600 python tools/wz_skip.py 512 512 20 20 8 2>&1 | filt > gpurun_out/r06_wz_skip_512.txt; echo wz_skip512 rc $?
# ---- 3: config 2 at 100 views ----
timeout 900 python -m pytest tests/test_gpu_modules.py -q -m gpu -k "test_full_view_count_scenes_against_the_oracle and cfg2_scannet_100v" 2>&1 | tail -5 > gpurun_out/r06_test_cfg2_100v.txt; echo test100v rc $?
timeout 600 python bench.py --workload cfg2_scannet_100v --no-cpu-baseline > gpurun_out/r06_bench_cfg2_100v.json 2>/dev/null; echo bench100v rc $?
# ---- 1c: per-layer A/B of the Winograd form with four scenes in flight (alternated pairs) ----
for rnd in 1 2; do
for deny in "" "512:128:20:20:8" "512:512:20:20:8" "512:128:20:20:8,512:512:20:20:8"; do
  SGC_WINOGRAD_Z_DENY="$deny" timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --steps 60 --warmup 15 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('deny=[$deny]', d['value'], 'sustained', d['sustained']['value'], 'self_check', d['self_check']['mismatching'])"
done; done > gpurun_out/r06_winograd_layers_ab.txt 2>&1; echo layers_ab rc $?
# ---- 7: power / clocks beside the headline, matrix pipe at the power limit ----
( for i in $(seq 1 40); do /opt/rocm/bin/rocm-smi --showclocks --showpower --showuse --json 2>/dev/null | head -c 1500; echo; sleep 0.5; done ) > gpurun_out/r06_smi_samples.txt &
SMI=$!
sleep 2
timeout 600 python bench.py --no-cpu-baseline --no-strict-fp32 --sustain 8 > gpurun_out/r06_bench_smi.json 2>/dev/null; echo smi-bench rc $?
wait $SMI
python - > gpurun_out/r06_smi_throughput.txt <<'PY'
import json
print(open("gpurun_out/r06_bench_smi.json").readline()[:400])
for ln in open("gpurun_out/r06_smi_samples.txt"):
    ln = ln.strip()
    if not ln.startswith("{"): continue
    try: d = json.loads(ln)
    except Exception: print(ln[:200]); continue
    c = d.get("card0", {})
    print({k: v for k, v in c.items() if any(s in k.lower() for s in ("sclk", "power", "use", "mclk", "fclk"))})
PY
rm -f gpurun_out/r06_smi_samples.txt
[ -x tools/probe/mfma_power ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_power.hip -o tools/probe/mfma_power 2>/dev/null
smi() { for i in $(seq 1 $1); do /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null | python3 -c "
import json,sys
try:
    c=json.load(sys.stdin)['card0']; print('   smi:', c.get('sclk clock speed:'), c.get('Current Socket Graphics Package Power (W)'), 'W')
except Exception as e: print('   smi: n/a')
"; sleep 0.4; done; }
for mode in "1 5 2" "0 5 2"; do
  echo "== mfma_power $mode (random data?, seconds, waves per SIMD)"
  smi 12 & S=$!
  timeout 60 tools/probe/mfma_power $mode
  wait $S
done > gpurun_out/r06_mfma_power.txt 2>&1
# ---- 1a / 2: SQ counters, separate passes, the program directly behind `--` ----
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/r06_counters_list.txt 2>&1
PA="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PB="GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
PC="SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA"
for part in a b c; do
  case $part in a) C=$PA;; b) C=$PB;; c) C=$PC;; esac
  rm -rf /tmp/pmc_w$part /tmp/pmc_d$part /tmp/pmc_g$part
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_w$part -- python3 $R/tools/conv_one.py 256 256 40 40 16 3 1 10 1 > /dev/null 2>&1; echo pmc wz $part rc $?
  timeout 200 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_d$part -- python3 $R/tools/conv_one.py 256 256 40 40 16 3 1 10 0 > /dev/null 2>&1; echo pmc direct $part rc $?
  SGC_TILE_CONFIGS="16,22,3,3,1,8,1,1,1,0" timeout 300 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_g$part -- python3 $R/tools/tile_bench.py cfg4 64x80 > /dev/null 2>&1; echo pmc gather $part rc $?
done
python3 - <<PY
import json, subprocess
for tag, d, pat in (("r06_pmc_conv_wz", "w", "conv3d_halo"), ("r06_pmc_conv_wz_out", "w", "winograd_z_out"), ("r06_pmc_conv_direct", "d", "conv3d_halo"), ("r06_pmc_gather_cm16", "g", "dfa3d_fwd_tile_kernel")):
    out = {}
    for part in "abc":
        try:
            out.update(json.loads(subprocess.run(["python3", "$R/tools/pmc_summary.py", f"/tmp/pmc_{d}{part}", pat, "2"], capture_output=True, text=True).stdout))
        except Exception as e:
            out["_error_" + part] = str(e)
    json.dump(out, open("$R/gpurun_out/" + tag + ".json", "w"), indent=1)
    print(tag, json.dumps(out))
PY
cd $R
cat gpurun_out/r06_base_cfg2_driver_cmd.json | cut -c1-300
cat gpurun_out/r06_wz_skip.txt gpurun_out/r06_wz_skip_512.txt gpurun_out/r06_test_cfg2_100v.txt gpurun_out/r06_winograd_layers_ab.txt
cut -c1-600 gpurun_out/r06_bench_cfg2_100v.json
tail -30 gpurun_out/r06_smi_throughput.txt; cat gpurun_out/r06_mfma_power.txt
