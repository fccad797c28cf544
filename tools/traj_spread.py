"""Spread of the engine's training trajectories against the oracle fixtures over repeated runs (one process, N replays): the table
behind the bounds of tests/test_gpu_zz_convergence.py.  Usage: python tools/traj_spread.py [N] > profiles/r04_traj_spread_<box>.txt"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)

import test_gpu_zz_convergence as T  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 5
import socket  # noqa: E402
print("host", socket.gethostname())
for fixture in ("aekl_traj_c1.json", "aekl_traj_thin.json"):
    for dtype in ("float32", "bfloat16"):
        rows = [T.replay_aekl(fixture, dtype) for _ in range(N)]
        print(f"== {fixture} {dtype}: per-step |got - want| / |want| of the adversarial terms, max over {N} runs (and run-to-run spread of got)")
        for k in ("recons", "spectral", "kl", "gen", "disc"):
            worst = max(max(r["rel"][k]) for r in rows)
            print(f"   {k:9s} worst over all steps and runs {worst:.3e}")
        for k in ("gen", "disc"):
            line = []
            for i in range(len(rows[0]["rel"][k])):
                gaps = [r["rel"][k][i] for r in rows]; gots = [r["got"][k][i] for r in rows]
                line.append(f"{i + 1}:{max(gaps):.1e}/{(max(gots) - min(gots)) / (abs(rows[0]['want'][k][i]) + 1e-12):.1e}")
            print(f"   {k} step:gap/spread " + " ".join(line))
for dtype in ("float32", "bfloat16"):
    w = [T.replay_ldm(dtype) for _ in range(N)]
    print(f"== ldm_traj_c2.json {dtype}: worst relative loss gap per run " + " ".join(f"{x:.2e}" for x in w))
    w = [T.replay_dm(dtype) for _ in range(N)]
    print(f"== dm_traj_c5.json {dtype}: worst relative loss gap per run " + " ".join(f"{x:.2e}" for x in w))
