"""CPU-only checks: the C-ABI library loads and exports every symbol include/vitk.h declares
(no compute calls), the ctypes table matches the header, the drop-in modules keep the reference's
state_dict keys, and host-side logic (LAI schedule, band rows, flop model) is right."""
import ctypes
import os
import re
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import golden_inputs as gi  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'vitk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(vitk_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from vit_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'run `python -m vit_amd._build` first'
    L = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 14
    for s in syms:
        assert hasattr(L, s), f'{s} declared in include/vitk.h but not exported'
    for s in _lib._SIGNATURES:
        assert s in syms, f'{s} bound by ctypes but not declared in include/vitk.h'
    assert L.vitk_version() >= 1


def test_header_arity_matches_ctypes_table():
    from vit_amd import _lib
    text = open(os.path.join(ROOT, 'include', 'vitk.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    for name, sig in _lib._SIGNATURES.items():
        m = re.search(r'\b' + name + r'\s*\(([^;]*?)\)\s*;', text, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ('', 'void') else len(args.split(','))
        assert n == len(sig[0]), (name, n, len(sig[0]))


def test_ops_fail_loudly_without_gpu_tensors():
    from vit_amd import ops
    from vit_amd._lib import VitkError
    with pytest.raises(VitkError):
        ops.gemm(torch.randn(8, 16), torch.randn(8, 16))


def test_state_dict_keys_match_reference_named_oracle():
    """The oracle's keys are the reference's (the golden tests load reference-keyed values into it)."""
    from vit_amd import model as M
    from vit_amd.disentangled_transformer.disentangled_transformer import DisentangledTransformer
    from vit_amd.timm_classes.timm_locality_aware_transformer import AttentionLAI, VisionTransformerLAI
    me = gi.LAI_MASK_EPOCHS

    def shapes(m):
        return {k: tuple(v.shape) for k, v in m.state_dict().items()}
    kw = dict(d_model=128, dropout=0.0, nhead=4, dim_feedforward=256, num_encoder_layers=2, normalize_before=True,
              pos_att_type='c2p|p2c', max_position_embeddings=128, relative_attention=True, lai=True, mask_epochs=me)
    mine = M.VisionTransformer(img_size=32, patch_size=4, num_classes=10, transformer=DisentangledTransformer(**kw))
    ref = R.RefVisionTransformer(img_size=32, patch_size=4, num_classes=10, transformer=R.RefDisentangledTransformer(**kw))
    assert shapes(mine) == shapes(ref)
    kw = dict(d_model=64, dropout=0.0, nhead=4, dim_feedforward=128, num_encoder_layers=2, normalize_before=True)
    mine = M.VisionTransformer(img_size=32, patch_size=4, num_classes=10, transformer=M.Transformer(**kw))
    ref = R.RefVisionTransformer(img_size=32, patch_size=4, num_classes=10, transformer=R.RefTransformer(**kw))
    assert shapes(mine) == shapes(ref)
    kw = dict(img_size=32, patch_size=4, num_classes=10, embed_dim=64, depth=2, num_heads=4, mask_epochs=me)
    assert shapes(VisionTransformerLAI(**kw)) == shapes(R.RefVisionTransformerLAI(**kw))
    assert shapes(AttentionLAI(64, 4, qkv_bias=True)) == shapes(R.RefAttentionLAI(64, 4, qkv_bias=True))


def test_lai_mask_and_schedule_match_golden():
    import hashlib
    import numpy as np
    from vit_amd.timm_classes.timm_locality_aware_transformer import get_attn_mask, lai_stage_width
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'lai_masks.npz'))
    for N in (65, 197, 577):
        for w in (8, 12, 18, 20):
            m = get_attn_mask(N, w)
            sha = np.frombuffer(hashlib.sha256(m.numpy().tobytes()).digest(), dtype=np.uint8)
            assert np.array_equal(sha, g[f'sha_{N}_{w}'])
    me = gi.LAI_MASK_EPOCHS
    assert [lai_stage_width(e, me) for e in (0, 9, 10, 19, 20, 29, 30, 39, 40, 99)] == \
        [8, 8, 12, 12, 18, 18, 20, 20, None, None]
    assert lai_stage_width(3, None) is None


def test_band_rows_and_relative_position():
    from vit_amd.disentangled_transformer.disentangled_transformer import band_rows
    from vit_amd.utils.ops import build_relative_position
    rel = torch.arange(2 * 8 * 3, dtype=torch.float32).view(16, 3)   # span 8
    N = 5
    band = band_rows(rel, N, 8)
    idx = R.dis_delta_index(N, 8)                                     # [N,N] clamp(i-j+S)
    for i in range(N):
        for j in range(N):
            assert torch.equal(band[i - j + N - 1], rel[idx[i, j]])
    big = band_rows(rel, 12, 8)                                       # clamp active: |delta| > span
    assert torch.equal(big[0], rel[0]) and torch.equal(big[-1], rel[15])
    rp = build_relative_position(4, 4)
    assert rp.shape == (1, 4, 4) and rp.dtype == torch.long and rp[0, 3, 1] == 2 and rp[0, 0, 2] == -2


def test_flop_model_matches_survey_table():
    sys.path.insert(0, ROOT)
    import bench
    from vit_amd.train import PRESETS
    f = bench.train_flops_per_image('disentangled_lai_attention', PRESETS['B'], 197, 16, 1000, 128, -1) / 1e9
    assert abs(f - 381.7) < 0.5                       # headline unmasked
    f = bench.train_flops_per_image('disentangled_lai_attention', PRESETS['B'], 197, 16, 1000, 128, 8) / 1e9
    assert abs(f - 360.7) < 0.5
    f = bench.train_flops_per_image('transformer_timm_lai', PRESETS['S'], 197, 16, 1000, 128, 20) / 1e9
    assert abs(f - 101.9) < 0.3


def test_gemm_async_epilogue_loads_are_never_spilled(tmp_path):
    """The GEMM's fast epilogue loads its side operands from inline asm and waits by hand; a register allocator that
    spills such a register before the wait corrupts whatever reuses it (tools/check_async_loads.py).  Compile the
    kernel file to assembly and check every kernel in it."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    # no skip: the library cannot be built without hipcc either, and a skipped check guards nothing
    assert os.path.exists(hipcc), 'hipcc is needed to compile gemm256.hip to assembly for the ISA check'
    from vit_amd import _build
    asm = tmp_path / 'gemm256.s'
    src = os.path.join(ROOT, 'vit_amd', 'csrc', 'gemm256.hip')
    subprocess.run([hipcc, *_build.FLAGS, '--cuda-device-only', '-S', src, '-o', str(asm)], check=True, capture_output=True)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_async_loads.py'), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]


_ASM_HEAD = """
_ZN4vitk4testEv:
\ts_load_dwordx2 s[0:1], s[4:5], 0x0
\t;;#ASMSTART
\tglobal_load_dwordx4 v[10:13], v[0:1], off
\t;;#ASMEND
"""
_ASM_TAIL = """
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_add_f32_e32 v20, v10, v11
\ts_endpgm
.Lfunc_end0:
"""


@pytest.mark.parametrize('body,rc,needle', [
    ('\tv_add_f32_e32 v5, v6, v7\n', 0, '0 violation(s), 0 note(s)'),                         # untouched: clean
    ('\tscratch_store_dword off, v11, off offset:16\n', 1, 'scratch_store'),                  # spill of unlanded data
    ('\tv_accvgpr_write_b32 a3, v12\n', 1, 'v_accvgpr_write'),                                # spill to an AGPR
    ('\tv_mov_b32_e32 v40, v13\n', 1, 'v_mov_b32'),                                           # VGPR-to-VGPR copy
    ('\tglobal_store_dwordx4 v[2:3], v[10:13], off\n', 1, 'global_store'),                    # stored before it landed
    ('\tv_mov_b32_e32 v12, 0\n', 0, '1 note(s)'),                                             # dead + reused: a note
    ('\ts_cbranch_scc0 .LBB0_2\n\ts_endpgm\n.LBB0_2:\n\tv_mov_b32_e32 v40, v10\n', 1, 'v_mov_b32'),   # past a first s_endpgm
    ('\tglobal_load_dword v30, v[0:1], off\n\ts_waitcnt vmcnt(1)\n\tv_mov_b32_e32 v40, v10\n', 0, '0 violation'),  # retired by count
])
def test_async_load_checker_rules(tmp_path, body, rc, needle):
    """tools/check_async_loads.py on hand-written assembly: what it must flag and what it must let pass."""
    import subprocess
    import sys
    f = tmp_path / 'k.s'
    f.write_text(_ASM_HEAD + body + _ASM_TAIL)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_async_loads.py'), str(f)], capture_output=True, text=True)
    assert r.returncode == rc, r.stdout
    assert needle in r.stdout, r.stdout


def test_async_load_checker_refuses_a_file_without_loads(tmp_path):
    import subprocess
    import sys
    f = tmp_path / 'k.s'
    f.write_text('_ZN4vitk4testEv:\n\tv_add_f32_e32 v1, v2, v3\n\ts_endpgm\n.Lfunc_end0:\n')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_async_loads.py'), str(f)], capture_output=True, text=True)
    assert r.returncode == 2, r.stdout


def test_dw_splitk_override_parses_tiles():
    """VITK_DW_SPLITK / bench.py --max-overlap: 'tiles:splitk,...' keyed on the number of 256 x 256 output tiles of the
    weight-gradient GEMM dW[out, in] = dy^T x; anything not listed keeps the library's choice (0)."""
    import torch
    from vit_amd import functional as VF
    old = VF._DW_SPLITK
    try:
        VF._DW_SPLITK = '16:2,64:2'
        rows = 8
        assert VF._dw_splitk(torch.empty(rows, 1024), torch.empty(rows, 1024)) == 2      # proj: 4 x 4 tiles
        assert VF._dw_splitk(torch.empty(rows, 4096), torch.empty(rows, 1024)) == 2      # fc1: 16 x 4
        assert VF._dw_splitk(torch.empty(rows, 1024), torch.empty(rows, 4096)) == 2      # fc2: 4 x 16
        assert VF._dw_splitk(torch.empty(rows, 3072), torch.empty(rows, 1024)) == 0      # qkv: 48 tiles, not listed
        assert VF._dw_splitk(torch.empty(rows, 1000), torch.empty(rows, 1025)) == 0      # 4 x 5 = 20 (ragged edges round up)
        VF._DW_SPLITK = ''
        assert VF._dw_splitk(torch.empty(rows, 1024), torch.empty(rows, 1024)) == 0
    finally:
        VF._DW_SPLITK = old
