#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/collect_profiles.sh on the GPU box) into the tracked
summaries under profiles/:
  <tag>_bench.json, <tag>_bench_under_rocprof.json   the bench lines
  <tag>_bench_kernel_stats.csv                        rocprofv3 --stats per-kernel summary
  <tag>_pmc_traffic.json                              HBM-side bytes per launch per kernel
FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch; on gfx950 FETCH_SIZE under-reports 16-B/lane
streams by 2x (MI355X_MICROARCH.md, HBM section) -- calibrated on quant_flat_vec, whose traffic is known exactly.
usage: python tools/summarise_profiles.py r1"""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"prof_{tag}"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit(f"no JSON line in {path}")


for name in ("bench", "bench_under_rocprof"):
    json.dump(last_json_line(os.path.join(src, name + ".json")), open(os.path.join(dst, f"{tag}_{name}.json"), "w"), indent=1)

stats = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))

# The --stats average of a kernel is over EVERY launch of the traced command, including the launches that run while the power controller is still
# climbing to its sustained clock (after the process' idle set-up and after every change of kernel).  The per-launch trace of the same pass shows
# it: windows of 50 consecutive launches of the dominant kernel, in launch order.  rocprofv3 stamps a back-to-back launch's start at the previous
# launch's end, so these durations also contain the ~2 us launch-to-launch gap that HIP events around a batch see as well.
trace = glob.glob(os.path.join(src, "trace", "**", "*kernel_trace.csv"), recursive=True)
if trace:
    rows = sorted(csv.DictReader(open(trace[0])), key=lambda r: int(r["Start_Timestamp"]))
    names = collections.Counter()   # the dominant kernel = the GEMM with the largest total time (not the most launches: the cfg1 block launches its small kernel 5000 times)
    for r in rows:
        if "asq::gemm_i8" in r["Kernel_Name"]:
            names[r["Kernel_Name"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    dom = names.most_common(1)[0][0]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if r["Kernel_Name"] == dom]
    live = last_json_line(os.path.join(src, "bench_under_rocprof.json"))["roofline"]
    with open(os.path.join(dst, f"{tag}_bench_kernel_trace_windows.txt"), "w") as f:
        f.write("# per-launch durations of the dominant kernel from the SAME rocprofv3 --kernel-trace --stats pass as %s_bench_kernel_stats.csv, in launch order,\n" % tag)
        f.write("# windows of 100 launches: mean / min / max us.  Order of the command: settle phase + warm-up + timed steps (3 launches per step, quantisers between),\n")
        f.write("# fused-QKV comparison (other kernel), then the dominant-kernel timing (200 warm launches + the settle time, then 20 x 50 event-timed launches = the LAST 1000).\n")
        f.write("# kernel: %s\n" % dom.split("(")[0])
        for j in range(0, len(d), 100):
            w = d[j:j + 100]
            f.write("%5d..%-5d  mean %6.2f  min %6.2f  max %6.2f\n" % (j, j + len(w) - 1, sum(w) / len(w), min(w), max(w)))
        f.write("all %d launches: mean %.2f us (the --stats AverageNs)\n" % (len(d), sum(d) / len(d)))
        f.write("last 1000 launches (the event-timed ones): mean %.2f us; bench.py's live HIP-event figure in this pass: avg %.2f us, min %.2f us\n" % (sum(d[-1000:]) / 1000, live["avg_us"], live["min_us"]))


def pmc_means(counter):
    f = glob.glob(os.path.join(src, f"pmc_{counter}", "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            acc[(r["Kernel_Name"], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


fetch, write = pmc_means("FETCH_SIZE"), pmc_means("WRITE_SIZE")
bench = last_json_line(os.path.join(src, "bench.json"))
import hashlib
out = {
    "library_sha256": hashlib.sha256(open(os.path.join(root, "autosmoothquant_amd", "libasq_hip.so"), "rb").read()).hexdigest(),   # bench.py reports these bytes only for THIS build of the library
    "command": "rocprofv3 --kernel-trace --pmc <C> --output-format csv -- python bench.py --no-cpu-baseline --no-cfg3 --steps 5 --warmup 2  (one pass per counter, C in {FETCH_SIZE, WRITE_SIZE})",
    "units": "rocprofv3 reports FETCH_SIZE/WRITE_SIZE in KiB per dispatch",
    "gfx950_correction": "FETCH_SIZE x2 (MI355X_MICROARCH.md 'HBM'); check: quant_flat_vec reads exactly M*K*2 B and writes M*K B",
    "workload": bench["config"]["workload"],
    "kernels": {},
}
for key in sorted(fetch, key=lambda k: -fetch[k][0]):
    name, grid = key
    if "asq::" not in name:
        continue
    short = name.split("(")[0].replace("void ", "")
    f_kib, n = fetch[key]
    w_kib = write.get(key, (0.0, 0))[0]
    e = {"grid_size": grid, "dispatches_sampled": n, "FETCH_SIZE_KiB_raw": round(f_kib, 1), "WRITE_SIZE_KiB_raw": round(w_kib, 1),
         "fetch_bytes_corrected": int(f_kib * 1024 * 2), "write_bytes": int(w_kib * 1024)}
    e["traffic_bytes_per_launch"] = e["fetch_bytes_corrected"] + e["write_bytes"]
    if ("gemm_i8_p8<" in short or "gemm_i8_p16<" in short) and grid == 256 * 512:  # 256 tiles of 256x256: the M=N=K=4096 launches of the default workload
        e["shape"] = [4096, 4096, 4096]
        e["algorithmic_bytes_per_launch"] = 4096 * 4096 * (1 + 1 + 2)
        e["traffic_over_algorithmic"] = round(e["traffic_bytes_per_launch"] / e["algorithmic_bytes_per_launch"], 2)
        e["note"] = ("every X panel is fetched by 2 XCD L2s and every W panel by 4 (each XCD owns a 4x8 block of 256x256 tiles), so memory-side reads are "
                     "~3x the 33.5 MB of operands; they are served by the 256 MiB Infinity Cache (FETCH_SIZE counts those hits)")
    out["kernels"][f"{short} grid={grid}"] = e
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_traffic.json"), "w"), indent=1)

# MFMA / LDS counters (each from its own pass), per kernel: mean over the sampled dispatches
extra = {}
for counter in ("MfmaUtil", "SQ_INSTS_VALU_MFMA_I8", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
    try:
        for (name, grid), (val, n) in pmc_means(counter).items():
            if "asq::gemm" in name:
                short = name.split("(")[0].replace("void ", "")
                extra.setdefault(f"{short} grid={grid}", {"dispatches_sampled": n})[counter] = round(val, 2)
    except (IndexError, KeyError):
        pass
if extra:
    json.dump({"command": "rocprofv3 --kernel-trace --pmc <C> -- python bench.py --no-cpu-baseline --no-cfg3 --steps 5 --warmup 2 (one pass per counter)",
               "notes": "MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * SIMD_NUM) in percent (rocprofv3 derived metric); "
                        "SQ_INSTS_VALU_MFMA_I8: 4096^3 needs 2*4096^3 / (2*16*16*64) = 4194304 wave-level v_mfma_i32_16x16x64_i8 (gemm_i8_p16; 2097152 of the 32x32x32 form for gemm_i8_p8)",
               "kernels": extra}, open(os.path.join(dst, f"{tag}_pmc_mfma_lds.json"), "w"), indent=1)
    print(json.dumps(extra, indent=1)[:2500])
# cfg3 (32-layer LLaMA-2-7B decoder forward, 65536 tokens): per-kernel share of the GPU time, reference composition and N1-fused
import re
for t in ("cfg3", "cfg3_fused"):
    fs = glob.glob(os.path.join(src, t, "**", "*kernel_stats.csv"), recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(os.path.join(dst, f"{tag}_{t}_kernel_stats.txt"), "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats -- python bench.py --workload llama7b_decoder_b32_s2048 {'--fuse-norm --fuse-qkv ' if 'fused' in t else ''}--steps 2 --warmup 1 --no-cpu-baseline\n")
        f.write("# 3 forwards + set-up + the dominant-kernel timing batches (gate GEMM at M = 8192: warm launches + the settle time + 20 x 50 timed launches, on images and on plain operands); share of total GPU kernel time\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
            f.write("%5.1f%%  calls %6s  avg %9.1f us  %s\n" % (float(r["TotalDurationNs"]) / tot * 100, r["Calls"], float(r["AverageNs"]) / 1e3, re.sub(r"\(.*", "", r["Name"])[:110]))
        f.write("total %.1f ms\n" % (tot / 1e6))
print(open(os.path.join(dst, f"{tag}_bench_kernel_stats.csv")).read())
print(json.dumps(out["kernels"], indent=1)[:3000])
