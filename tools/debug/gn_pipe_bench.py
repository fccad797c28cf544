"""A/B of the pipelined GroupNorm backward (gn_bwd_pipe_kernel) against the register-resident kernel at the LDM step's shapes, B = 256.
Inputs rotate through NSET buffer sets (> the 256 MB Infinity Cache) so that, as inside the step, x / dy come from HBM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import eegldm
from eegldm._lib import lib, ptr, check
ctx = eegldm.default_context(0)
B = int(os.environ.get("B", 256)); NSET = int(os.environ.get("NSET", 4))
shapes = [(768, 128), (384, 256), (192, 512), (768, 256), (384, 512), (192, 1024)]
def switch(**kv):
    for k, v in kv.items():
        if v is None: os.environ.pop(k, None)
        else: os.environ[k] = v
    lib.eegldm_debug_reload_env()
for (L, C) in shapes:
    R = B * L
    sets = []
    for s in range(NSET):
        x = torch.randn(R, C, device="cuda").bfloat16(); dy = torch.randn(R, C, device="cuda").bfloat16(); e = torch.randn(R, C, device="cuda").bfloat16()
        sets.append((x, dy, e, torch.empty_like(x)))
    ga = torch.ones(C, device="cuda"); be = torch.zeros(C, device="cuda"); st = torch.empty(B * 32 * 2, device="cuda"); dg = torch.zeros(C, device="cuda"); db = torch.zeros(C, device="cuda")
    y = torch.empty_like(sets[0][0])
    check(lib.eegldm_groupnorm_fwd(ctx.h, ptr(sets[0][0]), C, ptr(ga), ptr(be), ptr(y), C, ptr(st), B, L, C, 32, 1e-6, 1, 0, None, 0, 1))
    def run(i, with_e):
        x, dy, e, dx = sets[i % NSET]
        check(lib.eegldm_groupnorm_bwd(ctx.h, ptr(x), C, ptr(ga), ptr(be), ptr(st), ptr(dy), C, ptr(dx), C, ptr(dg), ptr(db), B, L, C, 32, 1, 0, ptr(e) if with_e else None, C, 1))
    def t(with_e, n=12):
        for i in range(NSET): run(i, with_e)
        ctx.sync(); ctx.timer_start()
        for i in range(n): run(i, with_e)
        return ctx.timer_stop_ms() / n * 1e3
    nb = R * C * 2
    out = []
    for name, env in (("resident", dict(EEGLDM_GN_NO_PIPE="1")), ("pipe", dict(EEGLDM_GN_NO_PIPE=None))):
        switch(**env)
        a, b_ = t(False), t(True)
        out.append(f"{name}: {a:6.1f} us ({3*nb/a/1e6:.2f} TB/s) | +addend {b_:6.1f} us ({4*nb/b_/1e6:.2f} TB/s)")
    print(f"L={L:4d} C={C:4d} {nb/1e6:4.0f} MB  " + "   ".join(out), flush=True)
