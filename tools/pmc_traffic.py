"""Summarise the rocprofv3 passes collected by tools/profile_r03.sh (round prefix: EEGLDM_PROFILE_ROUND, default r03; round 2 used tools/profile_r02.sh (all --output-format csv) into the files under profiles/:

    python tools/pmc_traffic.py <prof_dir> <tag>

  <prof_dir>/pmc_ldm_{FETCH_SIZE,WRITE_SIZE}   -> profiles/r02_pmc_hbm_traffic.json   HBM bytes per launch per GEMM class AND per HBM-bound
                                                                                       kernel family (gn_*, bn_*, dconv_*, adam, ...), with the
                                                                                       family's average duration -> achieved GB/s
  <prof_dir>/pmc_aekl_{FETCH_SIZE,WRITE_SIZE}  -> profiles/r02_pmc_aekl_step.json     the same per family + the whole AEKL/GAN step's bytes
  <prof_dir>/pmc_ldm_mfma                      -> profiles/r02_pmc_mfma_busy.json     MfmaUtil per GEMM class
  <prof_dir>/trace_*                           -> profiles/r02_*_kernel_stats_<tag>.txt per-kernel table (calls, total ms, average us, share)

gfx950 corrections (guides/MI355X_MICROARCH.md, HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in units of 1024 B;
FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 B, so it is DOUBLED; WRITE_SIZE is used as reported
(uncalibrated).  Every JSON carries `kernel_source_sha16` = the hash bench.py computes over csrc/, so bench.py can tell whether a
counter file belongs to the build it is timing."""
import csv
import glob
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = os.environ.get("EEGLDM_PROFILE_ROUND", "r06")        # prefix of the files written under profiles/
GEMM = {("1", "0", "3"): "conv3_fwd_implicit_gemm", ("1", "1", "3"): "conv3_dgrad_implicit_gemm", ("2", "1", "3"): "conv_wgrad_splitk_gemm",
        ("0", "0", "1"): "gemm_nt", ("0", "1", "1"): "gemm_nn", ("2", "1", "1"): "gemm_tn"}


def source_hash():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "synthetic-sleep-eeg-signal-generation-using-latent-diffusion-models_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def family(name):
    """kernel name -> (GEMM class | kernel family, is_gemm)"""
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.search(r"conv3_ws_kernel<(\w+)>", name)      # weight-stationary 3-tap conv (conv_ws.hip): forward / data gradient
    if m:
        return ("conv3_dgrad_implicit_gemm" if m.group(1) in ("true", "1") else "conv3_fwd_implicit_gemm"), True
    m = re.search(r"gemm_big_kernel<([^>]*)>", name)      # 192 x 256 tile (gemm_big.hip): <TAPS, KBLK, FLIP>, FLIP = data gradient
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return ("conv3_dgrad_implicit_gemm" if a[2] in ("true", "1") else "conv3_fwd_implicit_gemm"), True
    m = re.search(r"gemm_bigp_kernel<([^>]*)>", name)      # persistent form: <KBLK, FLIP, loader waves>
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        return ("conv3_dgrad_implicit_gemm" if a[1] in ("true", "1") else "conv3_fwd_implicit_gemm"), True
    m = re.search(r"gemm_big1p_kernel<(\w+),", name)      # persistent 1-tap form: <KBLK, loader waves>
    if m:
        return ("gemm_nn" if m.group(1) in ("true", "1") else "gemm_nt"), True
    m = re.search(r"gemm_big1_kernel<(\w+)>", name)      # 1 x 1 convs on the big tile: plain weights = forward (NT), K-blocked transposed copy = data gradient (the NN class)
    if m:
        return ("gemm_nn" if m.group(1) in ("true", "1") else "gemm_nt"), True
    m = re.search(r"gemm_kernel<([^>]*)>", name)
    if m:
        a = [x.strip() for x in m.group(1).split(",")]
        if a[0] == "float":
            return "gemm_f32", True
        return GEMM.get((a[1], a[2], a[3]), "gemm_other"), True
    m = re.match(r"([A-Za-z_0-9:]+)", name)
    base = m.group(1) if m else name
    base = base.split("::")[-1]
    return re.sub(r"_kernel$", "", base), False


def read_counter(d, counter):
    """-> {dispatch_id: (kernel_name, value summed over the counter's dimensions)}"""
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = (f, row.get("Dispatch_Id"))
            e = out.setdefault(k, [row["Kernel_Name"], 0.0])
            e[1] += float(row["Counter_Value"])
    return out


def read_trace(d):
    """-> list of (kernel_name, duration_ns) from a kernel trace"""
    out = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out.append((row["Kernel_Name"], int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    return out


def per_family(dfetch, dwrite):
    fe, wr = read_counter(dfetch, "FETCH_SIZE"), read_counter(dwrite, "WRITE_SIZE")
    dur = {}
    for n, t in read_trace(dfetch):
        e = dur.setdefault(family(n)[0], [0, 0]); e[0] += 1; e[1] += t
    agg = {}
    for src, key, mult in ((fe, "fetch", 2.0 * 1024), (wr, "write", 1024.0)):
        for (_f, _d), (name, val) in src.items():
            fam, is_gemm = family(name)
            e = agg.setdefault(fam, {"is_gemm": is_gemm, "n_fetch": 0, "n_write": 0, "fetch": 0.0, "write": 0.0})
            e["n_" + key] += 1; e[key] += val * mult
    res = {}
    tot_f = tot_w = 0.0
    for fam, e in agg.items():
        tot_f += e["fetch"]; tot_w += e["write"]
        f = e["fetch"] / max(1, e["n_fetch"]); w = e["write"] / max(1, e["n_write"])
        r = {"launches_sampled": e["n_fetch"], "fetch_bytes_per_launch": round(f), "write_bytes_per_launch": round(w), "hbm_bytes_per_launch": round(f + w)}
        if fam in dur and dur[fam][0]:
            us = dur[fam][1] / dur[fam][0] / 1e3
            r["avg_us_under_counters"] = round(us, 2); r["hbm_GBps_under_counters"] = round((f + w) / us / 1e3, 1)
        res[fam] = r
    return res, tot_f, tot_w


def kernel_table(d, path, header):
    rows = read_trace(d)
    if not rows:
        return
    agg = {}
    for n, t in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n.replace("void ", ""))
        m = re.match(r"(\w+)<(.*)>\(.*", n)
        k = (f"{m.group(1)}<{m.group(2)}>" if m else n.split("(")[0])[:110]
        e = agg.setdefault(k, [0, 0]); e[0] += 1; e[1] += t
    tot = sum(v[1] for v in agg.values())
    with open(path, "w") as f:
        f.write(header + f"\ntotal kernel time {tot/1e6:.2f} ms over {len(rows)} dispatches\n share   calls   total ms     avg us  kernel\n")
        for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if t / tot >= 0.002:
                f.write(f"{100*t/tot:6.2f} {c:7d} {t/1e6:10.3f} {t/c/1e3:10.2f}  {k}\n")
    print("wrote", path)


def main(prof, tag):
    P = os.path.join(ROOT, "profiles"); sha = source_hash()
    corr = "FETCH_SIZE x2 (gfx950: 128-byte requests tallied at 64 B), units of 1024 B; WRITE_SIZE as reported (uncalibrated)"
    fam, _tf, _tw = per_family(os.path.join(prof, "pmc_ldm_FETCH_SIZE"), os.path.join(prof, "pmc_ldm_WRITE_SIZE"))
    if fam:
        # whole-step traffic: every kernel of the run except the model set-up (state-dict copies, torch's init fills), divided by the number of
        # steps the run made (= Adam launches: one per step)
        n_steps = max(1, fam.get("adam", {}).get("launches_sampled", 1))
        setup = ("__amd_rocclr_copyBuffer", "vectorized_elementwise", "__amd_rocclr_fillBufferAligned")
        step_bytes = sum(v["hbm_bytes_per_launch"] * v["launches_sampled"] for k, v in fam.items() if k not in setup) / n_steps
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), tools/debug/quick_bench.py bfloat16 256 768 2 "
                             "(LDM train step B=256 bf16), EEGLDM_NO_SIDE_STREAM=1", "corrections": corr, "kernel_source_sha16": sha,
                   "steps_sampled": n_steps, "hbm_bytes_per_step": round(step_bytes),
                   "classes": {k: v for k, v in fam.items() if k in GEMM.values()},
                   "hbm_bound_families": {k: v for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])
                                          if k not in GEMM.values() and v["launches_sampled"] >= 2}},
                  open(os.path.join(P, f"{RND}_pmc_hbm_traffic.json"), "w"), indent=1)
        print("wrote", RND + "_pmc_hbm_traffic.json")
    fam, tf, tw = per_family(os.path.join(prof, "pmc_aekl_FETCH_SIZE"), os.path.join(prof, "pmc_aekl_WRITE_SIZE"))
    if fam:
        n_steps = 8        # tools/debug/aekl_bench.py: 3 warm-up + 5 timed steps, all identical
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/debug/aekl_bench.py 256 bfloat16 "
                             "(AEKL [2,2,4] + PatchDiscriminator GAN step, 8 steps; totals divided by 8; model set-up kernels included, <1 %)",
                   "corrections": corr, "kernel_source_sha16": sha, "hbm_bytes_per_step": round((tf + tw) / n_steps),
                   "fetch_bytes_per_step": round(tf / n_steps), "write_bytes_per_step": round(tw / n_steps),
                   "families": dict(sorted(fam.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"]))},
                  open(os.path.join(P, f"{RND}_pmc_aekl_step.json"), "w"), indent=1)
        print("wrote", RND + "_pmc_aekl_step.json")
    mu = read_counter(os.path.join(prof, "pmc_ldm_mfma"), "MfmaUtil")
    if mu:
        agg = {}
        for (_f, _d), (name, val) in mu.items():
            fam_, is_gemm = family(name)
            if is_gemm:
                e = agg.setdefault(fam_, [0, 0.0]); e[0] += 1; e[1] += val
        json.dump({"source": "rocprofv3 --pmc MfmaUtil --kernel-trace, tools/debug/quick_bench.py bfloat16 256 768 2, EEGLDM_NO_SIDE_STREAM=1",
                   "kernel_source_sha16": sha, "classes": {k: {"launches_sampled": n, "MfmaUtil_pct": round(v / n, 2)} for k, (n, v) in agg.items()}},
                  open(os.path.join(P, f"{RND}_pmc_mfma_busy.json"), "w"), indent=1)
        print("wrote", RND + "_pmc_mfma_busy.json")
    kernel_table(os.path.join(prof, "trace_ldm"), os.path.join(P, f"{RND}_ldm_step_bf16_B256_kernel_stats_{tag}.txt"),
                 "# EEGLDM_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -- python bench.py --no-parts --no-cpu-baseline --steps 7")
    kernel_table(os.path.join(prof, "trace_aekl"), os.path.join(P, f"{RND}_aekl_gan_step_bf16_B256_kernel_stats_{tag}.txt"),
                 "# EEGLDM_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -- python tools/debug/aekl_bench.py 256 bfloat16   (8 steps)")
    kernel_table(os.path.join(prof, "trace_parts"), os.path.join(P, f"{RND}_ddim50_and_pixel_dm_kernel_stats_{tag}.txt"),
                 "# EEGLDM_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -- python tools/debug/parts_bench.py   (DDIM-50 B=256 x2, B=1 x2; pixel DM 6 steps)")
    for src, dst in (("bench_ldm_line.json", f"{RND}_ldm_step_bf16_B256_rocprofv3.bench_line_{tag}.json.txt"),):
        s = os.path.join(prof, src)
        if os.path.exists(s) and os.path.getsize(s):
            open(os.path.join(P, dst), "w").write(open(s).read())
    st = glob.glob(os.path.join(prof, "trace_ldm", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        open(os.path.join(P, f"{RND}_ldm_step_bf16_B256_rocprofv3_kernel_stats_{tag}.csv"), "w").write(open(st[0]).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "v1")
