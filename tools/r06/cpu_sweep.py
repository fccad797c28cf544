"""Round-6 developer tool: thread / batch sweep of the CPU baseline (the oracle's LDM train step) with the OpenMP pool PINNED -- each point is
a fresh process started with OMP_NUM_THREADS = n, OMP_PROC_BIND = close, OMP_PLACES = cores and its affinity restricted to the first n
physical cores (`taskset`-style, os.sched_setaffinity before torch is imported), so that torch's intra-op pool neither migrates nor
straddles sockets unless n exceeds one socket.  bench.py's in-process sweep changes torch.set_num_threads only; VERDICT r5 (weak 12)
asked what a pinned sweep to 64 / 128 threads at B = 32 gives.

    python tools/r06/cpu_sweep.py                      # parent: runs the grid, prints one line per point
    python tools/r06/cpu_sweep.py --child B NT         # one point (internal)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def first_physical_cores(n):
    """logical CPU ids of the first n physical cores (one hyper-thread each), in (package, core) order"""
    seen, order = set(), []
    base = "/sys/devices/system/cpu"
    cpus = sorted(int(d[3:]) for d in os.listdir(base) if d.startswith("cpu") and d[3:].isdigit())
    for c in cpus:
        try:
            pk = int(open(f"{base}/cpu{c}/topology/physical_package_id").read())
            co = int(open(f"{base}/cpu{c}/topology/core_id").read())
        except OSError:
            continue
        if (pk, co) not in seen:
            seen.add((pk, co)); order.append((pk, co, c))
    order.sort()
    allowed = os.sched_getaffinity(0)
    ids = [c for _, _, c in order if c in allowed]
    return ids[:n]


def child(B, NT, pin, first=0):
    if pin:
        ids = first_physical_cores(first + NT)[first:]
        if ids:
            os.sched_setaffinity(0, ids)
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch
    torch.set_num_threads(NT)
    from oracle import losses as Ls, steps as S, unet as U
    from param_gen import gen_param, normal, timesteps
    cfg = dict(image_size=768, in_channels=1, out_channels=1, model_channels=128, num_res_blocks=2, attention_resolutions=[8, 4],
               channel_mult=[1, 2, 4], resblock_updown=True)
    sd = {k: torch.from_numpy(gen_param(42, k, s)) for k, s in U.unet_param_shapes(cfg).items()}
    acp = Ls.alphas_cumprod("linear_beta", 1000, 0.0015, 0.0195)
    lat, nz = torch.from_numpy(normal((B, 1, 768), seed=1)), torch.from_numpy(normal((B, 1, 768), seed=2))
    t = torch.from_numpy(timesteps(B, seed=3))
    st = {"sd": dict(sd), "opt": {}, "i": 0}
    def f():
        _l, grads, _ = S.ldm_train_step(st["sd"], cfg, acp, lat, nz, t)
        st["i"] += 1
        st["sd"] = S.adam_update(st["sd"], grads, st["opt"], 1e-4, st["i"])
    f()
    ts = []
    t_end = time.time() + float(os.environ.get("SWEEP_POINT_S", "12"))
    while len(ts) < 5 and (time.time() < t_end or not ts):
        t0 = time.time(); f(); ts.append(time.time() - t0)
    med = sorted(ts)[len(ts) // 2]
    print(json.dumps({"B": B, "threads": NT, "pinned": bool(pin), "windows_per_s": round(B / med, 2), "steps": len(ts),
                      "affinity": len(os.sched_getaffinity(0))}))


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--child":
        return child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]) if len(sys.argv) > 4 else 1, int(sys.argv[5]) if len(sys.argv) > 5 else 0)
    if len(sys.argv) >= 2 and sys.argv[1] == "--procs":      # P concurrent jobs of NT threads each on disjoint core slices: the host's aggregate
        B, NT = 32, 16
        for P in [int(a) for a in sys.argv[2:]] or [4, 8]:
            env = dict(os.environ, OMP_NUM_THREADS=str(NT), MKL_NUM_THREADS=str(NT), SWEEP_POINT_S="20")
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--child", str(B), str(NT), "1", str(i * NT)], env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for i in range(P)]
            rates = []
            for q in ps:
                out, _ = q.communicate(timeout=400)
                line = [l for l in out.splitlines() if l.startswith("{")]
                if line: rates.append(json.loads(line[-1])["windows_per_s"])
            print(json.dumps({"procs": P, "threads_each": NT, "B_each": B, "per_proc": rates, "aggregate_windows_per_s": round(sum(rates), 1)}), flush=True)
        return
    grid = [(8, 16, 0), (8, 16, 1), (8, 32, 1), (8, 64, 1), (32, 16, 1), (32, 32, 1), (32, 64, 1), (32, 128, 1), (32, 32, 0)]
    for B, NT, pin in grid:
        env = dict(os.environ, OMP_NUM_THREADS=str(NT), MKL_NUM_THREADS=str(NT))
        if pin:
            env.update(OMP_PROC_BIND="close", OMP_PLACES="cores")
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(B), str(NT), str(pin)], env=env, capture_output=True,
                               text=True, timeout=180)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else f"B {B} NT {NT} pin {pin}: rc {r.returncode} {r.stderr[-300:]}", flush=True)
        except subprocess.TimeoutExpired:
            print(f"B {B} NT {NT} pin {pin}: timeout", flush=True)


if __name__ == "__main__":
    main()
