#!/bin/bash
# One call that produces every profile committed under profiles/ for a round:
#   <tag>_bench_default.json(.err)        python bench.py (the driver's command)
#   <tag>_jasper_kernel_stats.csv         rocprofv3 --kernel-trace --stats of the Jasper step (bench streams)
#   <tag>_jasper_kernel_stats_serial.csv  the same with the weight-gradient stream folded into the main
#                                         stream (OS2S_WGRAD_STREAM=0): every kernel alone on the GPU
#   <tag>_pmc_bench_traffic.json          FETCH_SIZE / WRITE_SIZE per launch of the conv kernels (separate
#                                         --pmc passes, gfx950 correction) — bench.py reads the newest one
#   <tag>_pmc_mfma_busy.json              SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per
#                                         kernel (conv fwd+dgrad ping-pong, lockstep, wgrad)
#   <tag>_transformer_kernel_stats(.serial).csv, <tag>_transformer_pmc_mfma_busy.json, <tag>_gpu_idle_gaps.txt,
#   <tag>_{quartznet,ds2,tacotron,nmt}_kernel_stats.csv (5 steps each), <tag>_tacotron_decode_kernel_stats.csv,
#   <tag>_pmc_transformer_traffic.json, <tag>_pmc_frontend_traffic.json (bench.py reads the newest of each)
# Usage on the GPU box: bash tools/profile_round.sh r02
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profile_$TAG
mkdir -p $OUT
J="python bench.py --no-transformer --no-other-configs --no-cpu-baseline --no-kernel-timing"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ks -o jasper -- $J --steps 8 --warmup 3 > $OUT/ks.log 2>&1
cp $OUT/ks/jasper_kernel_stats.csv $OUT/${TAG}_jasper_kernel_stats.csv
OS2S_WGRAD_STREAM=0 timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kss -o jasper -- $J --steps 8 --warmup 3 > $OUT/kss.log 2>&1
cp $OUT/kss/jasper_kernel_stats.csv $OUT/${TAG}_jasper_kernel_stats_serial.csv
P="$J --steps 2 --warmup 1"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/f -o c -- $P > $OUT/f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/w -o c -- $P > $OUT/w.log 2>&1
# the same FETCH_SIZE pass with every kernel alone on the GPU (no L2 sharing with the other stream's kernels), and
# with the round-5 rank order of the weight-gradient kernels (conv1d_wgrad.xcd_order 0): what moved 357 -> 436 MB
OS2S_WGRAD_STREAM=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fs -o c -- $P > $OUT/fs.log 2>&1
OS2S_WGRAD_STREAM=0 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fx -o c -- $P --set-option conv1d_wgrad.xcd_order=0 > $OUT/fx.log 2>&1
OS2S_WGRAD_STREAM=0 timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/m -o c -- $P > $OUT/m.log 2>&1
python - <<PY
import csv, glob, json, collections
FAM = [("conv1d_pp_kernel", "conv fwd+dgrad, ping-pong tile (2 windows x 256 columns)"),
       ("conv1d_ppn_kernel", "conv fwd, narrow ping-pong tiles (2 / 3 windows x 128 columns)"), ("conv1d_igemm_grouped_kernel", "grouped 1x1, lockstep tile (round 6: the W C products of the dense-residual statistics)"),
       ("conv1x1_pp_kernel", "dense-residual GEMMs over the concatenated block inputs / gradients (256 x 256 ping-pong tile, row strides)"),
       ("conv1d_wgrad1x1_pp_kernel", "K = 1 TN GEMMs: dense-residual P_k and Gram matrices"),
       ("conv1d_igemm_kernel", "conv fwd+dgrad, lockstep tile"), ("conv1d_wgrad_pp_kernel", "wgrad, ping-pong tile"),
       ("conv1d_wgrad_kernel", "wgrad, lockstep tile")]
def fam(name):
    for k, _ in FAM:
        if k in name: return k
    return None
def collect(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = fam(r["Kernel_Name"])
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    return agg, cnt
fa, fc = collect("f"); wa, wc = collect("w"); ma, mc = collect("m")
fsa, fsc = collect("fs"); fxa, fxc = collect("fx")
per = {}
tf = tw = n = 0.0
for k, desc in FAM:
    if k not in fa: continue
    nf = fc[(k, "FETCH_SIZE")]; nw = wc[(k, "WRITE_SIZE")]
    f = fa[k]["FETCH_SIZE"] / max(nf, 1); w = wa[k]["WRITE_SIZE"] / max(nw, 1)
    per[k] = {"what": desc, "launches": nf, "FETCH_SIZE_KB_per_launch_raw": f, "WRITE_SIZE_KB_per_launch_raw": w,
              "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
              "fetch_bytes_per_launch_two_streams": 2.0 * f * 1024.0,
              "fetch_bytes_per_launch_alone_on_the_gpu": 2.0 * fsa[k]["FETCH_SIZE"] / max(fsc[(k, "FETCH_SIZE")], 1) * 1024.0,
              "fetch_bytes_per_launch_alone_round5_rank_order": 2.0 * fxa[k]["FETCH_SIZE"] / max(fxc[(k, "FETCH_SIZE")], 1) * 1024.0}
    if "wgrad" not in k and "grouped" not in k:
        tf += fa[k]["FETCH_SIZE"]; tw += wa[k]["WRITE_SIZE"]; n += nf
out = {"command": "$P", "kernel": "conv1d_pp_kernel + conv1d_ppn_kernel + conv1d_igemm_kernel + conv1x1_pp_kernel (the launches bench.py's roofline block times: fwd + dgrad)",
       "launches": n, "FETCH_SIZE_KB_per_launch_raw": tf / max(n, 1), "WRITE_SIZE_KB_per_launch_raw": tw / max(n, 1),
       "correction": "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE as reported (MI355X_MICROARCH.md)",
       "hbm_bytes_per_launch": (2.0 * tf / max(n, 1) + tw / max(n, 1)) * 1024.0, "per_kernel": per}
json.dump(out, open("$OUT/${TAG}_pmc_bench_traffic.json", "w"), indent=1)
busy = {"command": "OS2S_WGRAD_STREAM=0 $P", "note": "per-launch means; SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs: mfma_duty_cycle = MFMA_BUSY / (128 x GRBM_GUI_ACTIVE)", "per_kernel": {}}
for k, desc in FAM:
    if k not in ma: continue
    e = {c: ma[k][c] / mc[(k, c)] for c in ma[k]}
    e["launches"] = mc[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]
    if e.get("GRBM_GUI_ACTIVE"):
        e["mfma_duty_cycle"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * e["GRBM_GUI_ACTIVE"])   # GRBM summed over 8 XCDs
    e["what"] = desc
    busy["per_kernel"][k] = e
json.dump(busy, open("$OUT/${TAG}_pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes_per_launch"]) for k, v in per.items()}))
print(json.dumps({k: v.get("mfma_duty_cycle") for k, v in busy["per_kernel"].items()}))
PY
# ---- the other configurations: kernel stats of 8 steps each (serial streams: every kernel alone) ----
T="python bench.py --only-transformer --steps 5 --warmup 3"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tks -o tr -- $T > $OUT/tks.log 2>&1
cp $OUT/tks/tr_kernel_stats.csv $OUT/${TAG}_transformer_kernel_stats.csv
python tools/trace_gaps.py $OUT/tks/tr_kernel_trace.csv > $OUT/${TAG}_gpu_idle_gaps.txt 2>&1
python tools/trace_gaps.py $OUT/ks/jasper_kernel_trace.csv >> $OUT/${TAG}_gpu_idle_gaps.txt 2>&1
OS2S_DENSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/tkss -o tr -- $T > $OUT/tkss.log 2>&1
cp $OUT/tkss/tr_kernel_stats.csv $OUT/${TAG}_transformer_kernel_stats_serial.csv
OS2S_DENSE_WGRAD_STREAM=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $OUT/tm -o c -- python bench.py --only-transformer --steps 2 --warmup 1 > $OUT/tm.log 2>&1
python - <<PY
import csv, glob, json, collections
FAM = [("gemm_pp_kernel", "Dense forward / data gradient, 256x256 ping-pong tile"),
       ("conv1d_wgrad1x1_pp_kernel", "Dense weight gradient, 256x256 ping-pong tile"),
       ("conv1d_wgrad_kernel", "Dense weight gradient, lockstep tile (1024 x 1024 outputs)"),
       ("attn_bwd_kernel", "attention backward"), ("attn_fwd_kernel", "attention forward")]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("$OUT/tm/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = next((k for k, _ in FAM if k in r["Kernel_Name"]), None)
        if k is None: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
busy = {"command": "OS2S_DENSE_WGRAD_STREAM=0 python bench.py --only-transformer --steps 2 --warmup 1",
        "note": "per-launch means; mfma_duty_cycle = SQ_VALU_MFMA_BUSY_CYCLES / (128 x GRBM_GUI_ACTIVE)", "per_kernel": {}}
for k, desc in FAM:
    if k not in agg: continue
    e = {c: agg[k][c] / cnt[(k, c)] for c in agg[k]}
    e["launches"] = cnt[(k, "SQ_VALU_MFMA_BUSY_CYCLES")]
    if e.get("GRBM_GUI_ACTIVE"): e["mfma_duty_cycle"] = e["SQ_VALU_MFMA_BUSY_CYCLES"] / (128.0 * e["GRBM_GUI_ACTIVE"])
    e["what"] = desc
    busy["per_kernel"][k] = e
json.dump(busy, open("$OUT/${TAG}_transformer_pmc_mfma_busy.json", "w"), indent=1)
print(json.dumps({k: v.get("mfma_duty_cycle") for k, v in busy["per_kernel"].items()}))
PY
for m in quartznet ds2 tacotron nmt; do
  OS2S_WGRAD_STREAM=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/o_$m -o k -- python bench.py --only-$m --steps 3 --warmup 2 > $OUT/o_$m.log 2>&1
  cp $OUT/o_$m/k_kernel_stats.csv $OUT/${TAG}_${m}_kernel_stats.csv
done
# ---- free-running Tacotron2 decode (BASELINE configs[4]): kernel stats of the step kernels ----
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/o_dec -o k -- python bench.py --only-tacotron-decode --decode-steps 400 > $OUT/o_dec.log 2>&1
cp $OUT/o_dec/k_kernel_stats.csv $OUT/${TAG}_tacotron_decode_kernel_stats.csv
# ---- HBM traffic of the Transformer GEMM kernel and of the front end (separate --pmc passes) ----
TP="python bench.py --only-transformer --steps 2 --warmup 1"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/tf -o c -- $TP > $OUT/tf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/tw -o c -- $TP > $OUT/tw.log 2>&1
FP="python bench.py --only-frontend"
export OS2S_FRONTEND_NO_SATURATED=1     # the bench batch only: 23 calls (3 warm-up + 20 timed)
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/ff -o c -- $FP > $OUT/ff.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/fw -o c -- $FP > $OUT/fw.log 2>&1
unset OS2S_FRONTEND_NO_SATURATED
python - <<PY
import csv, glob, json, collections
def collect(sub, match):
    tot = collections.Counter(); n = collections.Counter()
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((m for m in match if m in r["Kernel_Name"]), None)
            if k is None: continue
            tot[k] += float(r["Counter_Value"]); n[k] += 1
    return tot, n
corr = "gfx950: FETCH_SIZE counts wide coalesced reads at 1/2 -> doubled; WRITE_SIZE as reported (MI355X_MICROARCH.md); both in KB"
f, fn = collect("tf", ["gemm_pp_kernel"]); w, wn = collect("tw", ["gemm_pp_kernel"])
if fn["gemm_pp_kernel"]:
    fk, wk = f["gemm_pp_kernel"] / fn["gemm_pp_kernel"], w["gemm_pp_kernel"] / max(wn["gemm_pp_kernel"], 1)
    json.dump({"command": "$TP", "kernel": "gemm_pp_kernel", "launches": fn["gemm_pp_kernel"], "FETCH_SIZE_KB_per_launch_raw": fk,
               "WRITE_SIZE_KB_per_launch_raw": wk, "correction": corr, "hbm_bytes_per_launch": (2.0 * fk + wk) * 1024.0},
              open("$OUT/${TAG}_pmc_transformer_traffic.json", "w"), indent=1)
names = ["logmel_frames_kernel", "logmel_normalize_kernel", "absmax_kernel", "logmel_stats_kernel"]
f, fn = collect("ff", names); w, wn = collect("fw", names)
if fn[names[0]]:
    # bench_frontend at the bench batch only (OS2S_FRONTEND_NO_SATURATED): 3 warm-up + 20 timed calls
    calls = fn[names[0]]
    per = {k: {"launches": fn[k], "FETCH_SIZE_KB_total": f[k], "WRITE_SIZE_KB_total": w[k],
               "hbm_bytes_per_batch": (2.0 * f[k] + w[k]) * 1024.0 / max(calls, 1)} for k in names if fn[k]}
    json.dump({"command": "OS2S_FRONTEND_NO_SATURATED=1 $FP", "kernels": per, "correction": corr,
               "note": "B = 32 bench batch, %d calls of the stage; hbm_bytes_per_launch = all kernels of ONE call" % calls,
               "hbm_bytes_per_launch": sum((2.0 * f[k] + w[k]) for k in names) * 1024.0 / max(calls, 1)},
              open("$OUT/${TAG}_pmc_frontend_traffic.json", "w"), indent=1)
PY
cp $OUT/${TAG}_pmc_transformer_traffic.json $OUT/${TAG}_pmc_frontend_traffic.json profiles/ 2>/dev/null
# ---- the driver's command, AFTER every PMC pass: bench.py reads roofline.traffic from the newest
#      profiles/*_pmc_bench_traffic.json, which must be the one committed next to its line ----
cp $OUT/${TAG}_pmc_bench_traffic.json profiles/${TAG}_pmc_bench_traffic.json
timeout 900 python bench.py > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
tail -c 600 $OUT/${TAG}_bench_default.json; echo
ls -la $OUT/*.json $OUT/*.csv $OUT/*.txt
