"""Summarise rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per pass, --output-format csv) into HBM bytes
per launch for each GEMM kernel class.  Usage: python tools/pmc_traffic.py <dir_fetch> <dir_write> <out.json>

gfx950 corrections (guides/MI355X_MICROARCH.md, HBM section): the counters are in KiB-like units of 1024 B as reported by
rocprofv3; FETCH_SIZE tallies 128-byte requests of wide coalesced reads at 64 B, so it is doubled; WRITE_SIZE is used as
reported (uncalibrated -- treat as a lower bound)."""
import csv, glob, json, os, re, sys

CLASSES = {("1", "0", "3"): "conv3_fwd_implicit_gemm", ("1", "1", "3"): "conv3_dgrad_implicit_gemm", ("2", "1", "3"): "conv_wgrad_splitk_gemm",
           ("0", "0", "1"): "gemm_nt", ("0", "1", "1"): "gemm_nn", ("2", "1", "1"): "gemm_tn"}


def collect(d, counter):
    out = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            m = re.search(r"gemm_kernel<([^>]*)>", name)
            if not m:
                continue
            a = [x.strip() for x in m.group(1).split(",")]
            if a[0] not in ("unsigned short", "__hip_bfloat16", "bf16_t"):
                pass
            cls = CLASSES.get((a[1], a[2], a[3]))
            if not cls or "float" == a[0]:
                continue
            e = out.setdefault(cls, [0, 0.0])
            e[0] += 1; e[1] += float(row["Counter_Value"])
    return out


def main(dfetch, dwrite, outp):
    fe, wr = collect(dfetch, "FETCH_SIZE"), collect(dwrite, "WRITE_SIZE")
    res = {}
    for cls in sorted(set(fe) | set(wr)):
        nf, vf = fe.get(cls, [0, 0.0]); nw, vw = wr.get(cls, [0, 0.0])
        fetch = 2.0 * vf * 1024 / max(nf, 1); write = vw * 1024 / max(nw, 1)
        res[cls] = {"launches_sampled": nf, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
                    "hbm_bytes_per_launch": fetch + write}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), quick_bench bf16 B=256 L=768",
               "corrections": "FETCH_SIZE x2 (gfx950 128-byte requests tallied at 64 B); units of 1024 B; WRITE_SIZE uncalibrated",
               "classes": res}, open(outp, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main(*sys.argv[1:4])
