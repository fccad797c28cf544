"""-m gpu: the AutoencoderKL / PatchDiscriminator primitives SURVEY 8b lists at operator granularity (VERDICT r4 "missing" item 3), through the
C ABI, against torch's fp32 ops and autograd on the same (bf16-rounded where the engine stores bf16) inputs:
  eegldm_batchnorm_lrelu_{fwd,bwd}   nn.BatchNorm1d (train: batch statistics + running-statistics update; eval) + LeakyReLU(0.2)
                                     -- MONAI PatchDiscriminator layers, config/config_aekl_eeg.yaml:30-40; twin src/models/discriminator.py:47-66
  eegldm_kl_reparam_{fwd,bwd}        AutoencoderKL.sampling + the KL of src/train_autoencoderkl.py:210-211 (clamp(-30, 20) of ae_kl.py:262-264)
Until round 5 both were parity-tested at model level only."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from param_gen import normal  # noqa: E402


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
@pytest.mark.parametrize("shape", [(3, 64, 256), (2, 128, 192), (4, 512, 96), (2, 6, 50)])
def test_batchnorm_leakyrelu_train_eval_and_backward(shape, dtype_name):
    import gpu_util as G
    c = G.ctx(); dt = G.F32 if dtype_name == "float32" else G.BF16
    B, C, L = shape
    x = torch.from_numpy(normal((B, C, L), seed=3)) * 1.5 + 0.4
    dy = torch.from_numpy(normal((B, C, L), seed=4))
    if dt == G.BF16:
        x = x.bfloat16().float(); dy = dy.bfloat16().float()
    gamma = 1.0 + 0.2 * torch.from_numpy(normal((C,), seed=5)); beta = 0.1 * torch.from_numpy(normal((C,), seed=6))
    rm0 = 0.05 * torch.from_numpy(normal((C,), seed=7)); rv0 = 1.0 + 0.1 * torch.from_numpy(normal((C,), seed=8)).abs()
    bn = torch.nn.BatchNorm1d(C)
    with torch.no_grad():
        bn.weight.copy_(gamma); bn.bias.copy_(beta); bn.running_mean.copy_(rm0); bn.running_var.copy_(rv0)
    xr = x.clone().requires_grad_(True)
    bn.train()
    ref = F.leaky_relu(bn(xr), 0.2)
    (ref * dy).sum().backward()
    xd, dyd = G.nlc(x, dt), G.nlc(dy, dt)
    gd, bd = gamma.to(G.DEV), beta.to(G.DEV)
    rm, rv, nbt = rm0.to(G.DEV), rv0.to(G.DEV), torch.zeros(1, device=G.DEV)
    st = torch.empty(C, 2, device=G.DEV); yd = torch.full_like(xd, float("nan"))
    G.check(G.lib.eegldm_batchnorm_lrelu_fwd(c.h, G.ptr(xd), C, G.ptr(gd), G.ptr(bd), G.ptr(st), G.ptr(rm), G.ptr(rv), G.ptr(nbt), G.ptr(yd), C,
                                             B * L, C, 0.2, 1, dt))
    G.assert_close(G.ncl(yd, B, L), ref.detach(), **G.TOL[dt], name="y (train)")
    assert torch.allclose(rm.cpu(), bn.running_mean, rtol=1e-5, atol=1e-6) and torch.allclose(rv.cpu(), bn.running_var, rtol=1e-5, atol=1e-6)
    assert float(nbt) == float(bn.num_batches_tracked) == 1.0
    mean = x.mean(dim=(0, 2)); var = x.var(dim=(0, 2), unbiased=False)
    assert torch.allclose(st[:, 0].cpu(), mean, rtol=1e-5, atol=1e-6) and torch.allclose(st[:, 1].cpu(), (var + 1e-5).rsqrt(), rtol=1e-5, atol=0)
    dxd = torch.full_like(xd, float("nan")); dg = torch.zeros(C, device=G.DEV); db = torch.zeros(C, device=G.DEV)
    G.check(G.lib.eegldm_batchnorm_lrelu_bwd(c.h, G.ptr(xd), C, G.ptr(gd), G.ptr(bd), G.ptr(st), G.ptr(dyd), C, G.ptr(dxd), C, G.ptr(dg), G.ptr(db),
                                             B * L, C, 0.2, dt))
    G.assert_close(G.ncl(dxd, B, L), xr.grad, **G.GTOL[dt], name="dx")
    scale = float(bn.weight.grad.abs().max())
    assert float((dg.cpu() - bn.weight.grad).abs().max()) < 2e-4 * scale + 1e-5 and float((db.cpu() - bn.bias.grad).abs().max()) < 2e-4 * float(bn.bias.grad.abs().max()) + 1e-5
    # eval mode: running statistics, nothing updated
    bn.eval()
    ref_e = F.leaky_relu(bn(x), 0.2)
    rm_before = rm.clone()
    G.check(G.lib.eegldm_batchnorm_lrelu_fwd(c.h, G.ptr(xd), C, G.ptr(gd), G.ptr(bd), G.ptr(st), G.ptr(rm), G.ptr(rv), G.ptr(nbt), G.ptr(yd), C,
                                             B * L, C, 0.2, 0, dt))
    G.assert_close(G.ncl(yd, B, L), ref_e.detach(), **G.TOL[dt], name="y (eval)")
    assert torch.equal(rm, rm_before) and float(nbt) == 1.0
    # gamma == NULL: plain LeakyReLU
    G.check(G.lib.eegldm_batchnorm_lrelu_fwd(c.h, G.ptr(xd), C, None, None, None, None, None, None, G.ptr(yd), C, B * L, C, 0.2, 1, dt))
    G.assert_close(G.ncl(yd, B, L), F.leaky_relu(x, 0.2), **G.TOL[dt], name="plain LeakyReLU")


@pytest.mark.parametrize("dtype_name", ["float32", "bfloat16"])
@pytest.mark.parametrize("shape", [(4, 1, 768), (3, 3, 64)])
def test_kl_reparameterisation_forward_and_backward(shape, dtype_name):
    import gpu_util as G
    c = G.ctx(); dt = G.F32 if dtype_name == "float32" else G.BF16
    B, lat, Ll = shape
    mu = torch.from_numpy(normal((B, lat, Ll), seed=11)); lv = torch.from_numpy(normal((B, lat, Ll), seed=12)) * 1.3
    lv[0, 0, :4] = torch.tensor([-40.0, -30.0, 25.0, 20.0])                      # the clamp and its gradient gate (ae_kl.py:262-264)
    eps = torch.from_numpy(normal((B, lat, Ll), seed=13)); dz = torch.from_numpy(normal((B, lat, Ll), seed=14))
    if dt == G.BF16:
        mu, lv, dz = mu.bfloat16().float(), lv.bfloat16().float(), dz.bfloat16().float()
    klw = 0.7
    mur, lvr = mu.clone().requires_grad_(True), lv.clone().requires_grad_(True)
    sg = torch.exp(torch.clamp(lvr, -30.0, 20.0) / 2)
    z = mur + eps * sg
    kl = 0.5 * torch.sum(mur.pow(2) + sg.pow(2) - torch.log(sg.pow(2)) - 1, dim=[1])
    kl = torch.sum(kl) / kl.shape[0]                                               # train_autoencoderkl.py:210-211
    ((z * dz).sum() + klw * kl).backward()
    n = B * lat * Ll
    mud, lvd, dzd = G.nlc(mu, dt), G.nlc(lv, dt), G.nlc(dz, dt)
    epsd = eps.permute(0, 2, 1).contiguous().to(G.DEV)                           # fp32, NLC like the tensors it multiplies
    zd = torch.full_like(mud, float("nan")); sgd = torch.empty(B * Ll, lat, device=G.DEV); kld = torch.zeros(1, device=G.DEV)
    G.check(G.lib.eegldm_kl_reparam_fwd(c.h, G.ptr(mud), G.ptr(lvd), G.ptr(epsd), G.ptr(zd), G.ptr(sgd), G.ptr(kld), n, B, dt))
    G.assert_close(G.ncl(zd, B, Ll), z.detach(), **G.TOL[dt], name="z")
    assert torch.allclose(sgd.reshape(B, Ll, lat).permute(0, 2, 1).cpu(), sg.detach(), rtol=2e-6, atol=0)
    assert abs(float(kld) - float(kl)) < 2e-6 * abs(float(kl))
    dmud = torch.full_like(mud, float("nan")); dlvd = torch.full_like(mud, float("nan"))
    G.check(G.lib.eegldm_kl_reparam_bwd(c.h, G.ptr(mud), G.ptr(lvd), G.ptr(epsd), G.ptr(sgd), G.ptr(dzd), G.ptr(dmud), G.ptr(dlvd), n, klw / B, dt))
    G.assert_close(G.ncl(dmud, B, Ll), mur.grad, **G.GTOL[dt], name="dmu")
    # at the clamp edges torch's clamp passes the gradient at == the bound (-30, 20); the reference never sits exactly there, and the
    # engine gates with strict inequalities: compare away from the four planted edge values
    got = G.ncl(dlvd, B, Ll).cpu(); want = lvr.grad.clone()
    got[0, 0, :4] = 0; wedge = want[0, 0, :4].clone(); want[0, 0, :4] = 0
    G.assert_close(got, want, **G.GTOL[dt], name="dlog_var")
    assert float(wedge[0]) == 0.0 and float(wedge[2]) == 0.0 and float(G.ncl(dlvd, B, Ll)[0, 0, 0]) == 0.0 and float(G.ncl(dlvd, B, Ll)[0, 0, 2]) == 0.0
