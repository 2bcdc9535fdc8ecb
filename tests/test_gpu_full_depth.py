"""Full-depth checks at the headline model size (24 layers x 1024 x 4096 x 16 heads, 197 tokens): everything else in the
suite that compares numbers runs 2 layers or a small width."""
import io
import json
import os
import sys
from contextlib import redirect_stdout

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_full_depth_bf16_tracks_fp32_for_five_steps():
    """ViT-B/16-224 LAI + disentangled (BASELINE.json's config), batch 16, LAI stage w = 20: the bf16 compute path and
    the repo's fp32 path (exact f32 MFMA, the one the 1e-5 parity tests pin to the oracle) start from identical weights
    and take five AdamW steps on the same batch.  Measured: fp32 6.846 5.394 4.606 8.750 4.121, bf16 6.847 5.401 4.615
    8.778 4.122 -- relative gaps 2e-4 .. 3.2e-3, the largest on the step where both overshoot.  The bar is 1e-2
    relative; a wrong kernel at depth shows up as O(0.1-1) or as a loss that does not fall."""
    from vit_amd.optim import AdamW
    from vit_amd.train import build_model, train_step
    me = [10, 20, 30, 40]
    torch.manual_seed(0)
    net32 = build_model('disentangled_lai_attention', 'B', mask_epochs=me, dtype=torch.float32)
    net16 = build_model('disentangled_lai_attention', 'B', mask_epochs=me, dtype=torch.bfloat16)
    net16.load_state_dict(net32.state_dict())
    g = torch.Generator(device='cuda').manual_seed(3)
    x = torch.randn(16, 3, 224, 224, device='cuda', generator=g)
    y = torch.randint(0, 1000, (16,), device='cuda', generator=g)
    o32, o16 = AdamW(net32.parameters(), lr=1e-4), AdamW(net16.parameters(), lr=1e-4)
    l32, l16 = [], []
    for _ in range(5):
        l32.append(float(train_step(net32, o32, x, y, 30)))
        l16.append(float(train_step(net16, o16, x, y, 30)))
    print('fp32', l32, 'bf16', l16)
    assert all(abs(a - b) < 1e-2 * max(1.0, abs(a)) for a, b in zip(l32, l16)), (l32, l16)
    assert l32[-1] < l32[0] - 0.5 and l16[-1] < l16[0] - 0.5, (l32, l16)     # one batch, five steps: both memorise it


def test_bench_with_forced_reducer():
    """bench.py --force-reducer: the headline step through the gradient reducer (flat slabs, direct-write slots, hooks,
    bucketed RCCL all-reduce) at world size 1 -- the code path of every N > 1 run, on the real model.  Each line comes from
    a process of its own, as the driver runs bench.py: the process group is made before anything has trained (RCCL's stream
    in the high-priority class, vit_amd.ddp.init_process_group), and a crash inside RCCL cannot take the pytest session
    with it (tests/test_gpu_ddp_nccl.py has the one abort seen)."""
    import subprocess
    lines = []
    for extra in ([], ['--force-reducer']):
        # three warm-up steps: the reducer builds its slabs on the first one and takes its direct-write slots from the second
        env = dict(os.environ, MASTER_PORT=str(29700 + os.getpid() % 200))
        r = subprocess.run([sys.executable, '-u', os.path.join(ROOT, 'bench.py'), '--steps', '4', '--warmup', '3', '--batch', '32',
                            '--no-cpu-baseline', '--no-isolated'] + extra, cwd=ROOT, env=env, capture_output=True, text=True,
                           timeout=280)
        assert r.returncode == 0, f'{extra}: rc {r.returncode}\n{r.stdout[-1000:]}\n{r.stderr[-3000:]}'
        lines.append(json.loads(r.stdout.strip().splitlines()[-1]))
    plain, red = lines
    assert red['n_gpus'] == 1 and red['steps'] == 4 and red['value'] > 0
    assert red['roofline']['kernel'].startswith('gemm')
    # same seed, same batch, seven steps each: the reducer changes where gradients are summed, not what they are
    assert abs(plain['loss'] - red['loss']) < 2e-2, (plain['loss'], red['loss'])
    assert red['value'] > 0.8 * plain['value'], (plain['value'], red['value'])    # slabs + hooks + 1-rank all-reduce stay cheap


@pytest.mark.parametrize('dtype,epoch', [(torch.float32, 30), (torch.bfloat16, 30), (torch.bfloat16, 45)],
                         ids=['f32-w20', 'bf16-w20', 'bf16-unmasked'])
def test_full_depth_against_the_oracle(dtype, epoch):
    """The headline model at FULL depth and width (24 layers x 1024 x 4096 x 16 heads, 224 px / 16 = 197 tokens, LAI stage
    w = 20) against the CPU oracle (oracle/ref_cpu.py, the restatement pinned to the reference's own outputs) on a batch of
    two: logits, label-smoothing loss, and the gradients of parameters from both ends and the middle of the stack.  In bf16
    this is the path of bench.py's headline line: attention_lai.hip, the extended-row layout, the fused layer node.
    Measured (max-abs error over max-abs reference): fp32 logits 2.0e-6, loss 6e-8, gradients <= 4.2e-6; bf16 logits
    1.25e-2, loss 1e-3, gradients <= 2.6e-2 (relative L2 <= 2.5e-2; the largest is the relative table, which every layer
    reads).  Bars: fp32 1e-5 on everything -- north_star's bar, at full depth; bf16 2e-2 on the logits and 4e-2 on the
    gradients (the documented 2x factor of the whole-model bf16 tests: 48 LayerNorms deep on rounded activations).
    epoch 45 is past the last LAI stage: the UNMASKED attention of the per-head kernels (attn_fwd_full_kernel /
    attn_bwd_full_kernel + attn_bwd_relds_kernel) inside the fused layer node, same bars."""
    import numpy as np
    sys.path.insert(0, ROOT)
    from oracle import ref_cpu as R
    from vit_amd.train import build_model, label_smoothing_ce
    me = [10, 20, 30, 40]
    torch.manual_seed(5)
    net = build_model('disentangled_lai_attention', 'B', mask_epochs=me, dtype=dtype)
    kw = dict(d_model=1024, dropout=0.0, nhead=16, dim_feedforward=4096, num_encoder_layers=24, normalize_before=True,
              pos_att_type='c2p|p2c', max_position_embeddings=1024, relative_attention=True, lai=True, mask_epochs=me)
    ref = R.RefVisionTransformer(img_size=224, patch_size=16, in_chans=3, num_classes=1000,
                                 transformer=R.RefDisentangledTransformer(**kw))
    ref.load_state_dict({k: v.detach().float().cpu() for k, v in net.state_dict().items()})
    g = torch.Generator().manual_seed(9)
    img = torch.randn(2, 3, 224, 224, generator=g)
    tgt = torch.tensor([3, 977])
    # epoch 30: stage w = 20; 45: no mask
    lo_r = ref(img, epoch)
    ls_r = R.label_smoothing_ce(lo_r, tgt)
    ls_r.backward()
    lo_g = net(img.cuda(), epoch)
    ls_g = label_smoothing_ce(lo_g, tgt.cuda())
    ls_g.backward()

    def maxerr(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    def l2err(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()

    f32 = dtype == torch.float32
    report = {'logits': maxerr(lo_g.float(), lo_r), 'loss': abs(ls_g.item() - ls_r.item()) / abs(ls_r.item())}
    assert report['logits'] < (1e-5 if f32 else 2e-2), report
    assert report['loss'] < (1e-5 if f32 else 1e-2), report
    rp = dict(ref.named_parameters())
    names = ['head.weight', 'patch_embed.proj.weight', 'transformer.encoder.rel_embeddings.weight',
             'transformer.encoder.layers.0.dis_attn.query_proj.weight', 'transformer.encoder.layers.0.dis_attn.key_proj.bias',
             'transformer.encoder.layers.11.dense_intermediate.weight', 'transformer.encoder.layers.23.dense_out.weight',
             'transformer.encoder.layers.23.layer_norm_out.weight', 'transformer.encoder.layers.12.dense_AttnOut.bias']
    gp = dict(net.named_parameters())
    for n in names:
        assert gp[n].grad is not None and rp[n].grad is not None, n
        report[n] = (maxerr(gp[n].grad.float(), rp[n].grad), l2err(gp[n].grad.float(), rp[n].grad))
    print(report)
    for n in names:
        mx, l2 = report[n]
        if f32:
            assert mx < 1e-5, (n, report[n])
        else:
            assert l2 < 4e-2 and mx < 4e-2, (n, report[n])
    # parameters the reference never trains get no gradient here either
    assert sorted(n for n, p in gp.items() if p.grad is None) == sorted(n for n, p in rp.items() if p.grad is None)


@pytest.mark.parametrize('epoch', [5, 25, 45], ids=['w8', 'w18', 'unmasked'])
def test_timm_lai_at_224px_against_the_oracle(epoch):
    """The timm LAI ViT at its real token count and head width (config 2's shape: 224 px / 16 = 197 tokens, 12 heads of 64,
    768 wide; four layers keep the CPU oracle short) in bf16 against the oracle: the plain (no relative tables) instances of
    the per-head attention kernels -- banded (attn_fwd_lai_kernel / attn_bwd_lai_kernel, BIAS = false) in the LAI stages,
    whole score rows (attn_fwd_full_kernel / attn_bwd_full_kernel, BIAS = false) past the last one -- inside the fused
    pre-LN block.  Bars as for the other whole-model bf16 tests: logits 2e-2, gradients 4e-2 (measured: logits 0.7-0.8e-2,
    worst gradient 1.3-1.4e-2 in all three stages)."""
    sys.path.insert(0, ROOT)
    from oracle import ref_cpu as R
    from vit_amd.train import build_model, label_smoothing_ce
    me = [10, 20, 30, 40]
    torch.manual_seed(7)
    net = build_model('transformer_timm_lai', 'S', mask_epochs=me, dtype=torch.bfloat16, depth=4)
    ref = R.RefVisionTransformerLAI(img_size=224, patch_size=16, num_classes=1000, embed_dim=768, depth=4, num_heads=12,
                                    mlp_ratio=4.0, mask_epochs=me)
    ref.load_state_dict({k: v.detach().float().cpu() for k, v in net.state_dict().items()})
    g = torch.Generator().manual_seed(11)
    img = torch.randn(3, 3, 224, 224, generator=g)
    tgt = torch.tensor([1, 500, 999])
    lo_r = ref(img, epoch)
    R.label_smoothing_ce(lo_r, tgt).backward()
    lo_g = net(img.cuda(), epoch)
    label_smoothing_ce(lo_g, tgt.cuda()).backward()

    def maxerr(a, b):
        a, b = a.detach().double().cpu(), b.detach().double().cpu()
        return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    report = {'logits': maxerr(lo_g.float(), lo_r)}
    rp = dict(ref.named_parameters())
    for n, q in net.named_parameters():
        report[n] = maxerr(q.grad.float(), rp[n].grad)
    worst = max(report.items(), key=lambda kv: kv[1] if kv[0] != 'logits' else 0.0)
    print({'logits': report['logits'], 'worst gradient': worst})
    assert report['logits'] < 2e-2, report['logits']
    assert worst[1] < 4e-2, worst
