"""CPU: what fp16 storage costs the SD2-depth UNet relative to fp32, established on the oracle alone.
This pins the yardstick used by the GPU parity test: an fp16 implementation cannot be closer to fp32 than
the weight-rounding floor, so the engine is compared with the fp16-storage restatement (<= 1e-3) and with
fp32 (<= 1.25x the restatement's own distance)."""
import torch
from oracle import unet_ref as U


def test_fp16_storage_floor():
    torch.manual_seed(1)
    cfg = U.tiny_config()
    ref = U.randomize_affine(U.UNet2DConditionModelRef(cfg)).eval()
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 5, 16, 16, generator=g)
    ctx = torch.randn(2, 7, cfg['cross_attention_dim'], generator=g)
    t = torch.tensor(981.0)
    with torch.no_grad():
        want = ref(x, t, ctx)['sample']
    rel = lambda a: ((a - want).norm() / want.norm()).item()
    full = rel(U.forward_fp16_storage(ref, x, t, ctx)['sample'])
    ident = lambda v: v
    weights_only = rel(U.forward_fp16_storage(ref, x, t, ctx, q=ident, q_res=ident)['sample'])          # fp16 weights, fp32 activations
    fp32_stream = rel(U.forward_fp16_storage(ref, x, t, ctx, q_res=ident)['sample'])                    # fp32 residual stream only
    exact = rel(U.forward_fp16_storage(ref, x, t, ctx, q=ident, q_res=ident, q_w=ident)['sample'])
    print(f"fp16 storage: full {full:.3e}, fp32 residual stream {fp32_stream:.3e}, weights only {weights_only:.3e}")
    assert exact < 1e-6                                   # the restatement with no rounding IS the fp32 forward
    assert 5e-4 < weights_only < full < 4e-3              # weight rounding alone already costs most of 1e-3
    assert weights_only < fp32_stream <= full * 1.05
