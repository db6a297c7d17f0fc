"""CPU: host-side logic of the product package and the C-ABI surface (no device calls)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib_path():
    from petr_amd import _C
    if not os.path.exists(_C.LIB_PATH):
        subprocess.run(['make', '-C', os.path.join(ROOT, 'petr_amd', 'csrc'), '-j', '8'], check=True)
    return _C.LIB_PATH


def test_cabi_exports_every_declared_symbol(lib_path):
    """libpetr_hip.so loads and exports every function include/petr_hip.h declares."""
    hdr = open(os.path.join(ROOT, 'include', 'petr_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(petr_[a-z0-9_]+)\s*\(', hdr))
    assert len(declared) >= 30
    lib = ctypes.CDLL(lib_path)
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, f'declared in the header but not exported: {missing}'
    from petr_amd import _C
    assert set(_C.EXPORTS) == declared, set(_C.EXPORTS) ^ declared
    lib.petr_version.restype = ctypes.c_int
    assert lib.petr_version() == 100


def test_product_library_reads_no_environment(lib_path):
    """"library holds no mutable global state".  The product build folds every tuning switch to its default
    (common.h petr_tune): libpetr_hip.so must not even import getenv (only -DPETR_TUNING_ENV diagnostic builds do)."""
    import subprocess
    syms = subprocess.run(['nm', '-D', '--undefined-only', lib_path], capture_output=True, text=True, check=True).stdout
    assert 'getenv' not in syms and 'setenv' not in syms


def test_ctypes_struct_sizes_match_header(lib_path, tmp_path):
    """sizeof of every argument struct as the C compiler sees it == the ctypes mirror."""
    from petr_amd import _C
    src = tmp_path / 'sz.c'
    names = ['petr_coords3d_args', 'petr_sine3d_args', 'petr_gemm_args', 'petr_layernorm_args',
             'petr_layernorm_bwd_args', 'petr_mha_fwd_args', 'petr_mha_bwd_args', 'petr_bbox_args', 'petr_head_config',
             'petr_head_layout_t', 'petr_head_io', 'petr_head_grads', 'petr_mha_bwd_bf16_args', 'petr_mha_fwd_bf16_args',
             'petr_decode_topk_args', 'petr_decode_args', 'petr_loss_args', 'petr_attn_out_ln_args', 'petr_ln_proj_args', 'petr_ln_bwd_proj_args', 'petr_ffn_fwd_args', 'petr_ffn_bwd_args', 'petr_wgrad_item', 'petr_branch_fwd_args', 'petr_branch_bwd_args', 'petr_task_heads_fwd_args', 'petr_task_heads_bwd_args']
    body = '\n'.join(f'printf("%zu\\n", sizeof({n}));' for n in names)
    src.write_text(f'#include <stdio.h>\n#include "petr_hip.h"\nint main(){{{body} return 0;}}')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    mirrors = [_C.Coords3dArgs, _C.Sine3dArgs, _C.GemmArgs, _C.LayerNormArgs, _C.LayerNormBwdArgs, _C.MhaFwdArgs,
               _C.MhaBwdArgs, _C.BboxArgs, _C.HeadConfig, _C.HeadLayout, _C.HeadIO, _C.HeadGrads, _C.MhaBwdBf16Args, _C.MhaFwdArgs,
               _C.DecodeTopkArgs, _C.DecodeArgs, _C.LossArgs, _C.AttnOutLnArgs, _C.LnProjArgs, _C.LnBwdProjArgs, _C.FfnFwdArgs, _C.FfnBwdArgs, _C.WgradItem, _C.BranchFwdArgs, _C.BranchBwdArgs, _C.TaskHeadsFwdArgs, _C.TaskHeadsBwdArgs]
    for n, s, m in zip(names, sizes, mirrors):
        assert ctypes.sizeof(m) == s, f'{n}: C says {s}, ctypes says {ctypes.sizeof(m)}'


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'petr_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), f'{f} imports the oracle'


def test_missing_library_fails_loudly(monkeypatch):
    from petr_amd import _C
    monkeypatch.setattr(_C, '_lib', None)
    monkeypatch.setattr(_C, 'LIB_PATH', '/nonexistent/libpetr_hip.so')
    with pytest.raises(_C.PetrHipError):
        _C.lib()


def test_head_rejects_cpu_tensors(lib_path):
    import petr_amd
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=16))
    metas = [{'pad_shape': [(128, 192, 3)] * 2, 'img_shape': [(128, 192, 3)] * 2, 'lidar2img': [np.eye(4)] * 2}]
    with pytest.raises(petr_amd._C.PetrHipError):
        head([torch.zeros(1, 2, 256, 4, 6)], metas)


@pytest.mark.parametrize('pad,img,hw', [((128, 192), (100, 150), (4, 6)), ((512, 1408), (512, 1408), (16, 44)),
                                        ((320, 800), (300, 790), (20, 50)), ((640, 1600), (1, 1), (40, 100)),
                                        ((97, 131), (50, 77), (7, 9))])
def test_padding_mask_closed_form_equals_interpolate(pad, img, hw):
    from oracle import petr_oracle as O
    from petr_amd.petr_head import padding_mask_closed_form
    metas = [{'pad_shape': [(pad[0], pad[1], 3)] * 3, 'img_shape': [(img[0], img[1], 3), (pad[0], pad[1], 3), (img[0], pad[1], 3)]}
             for _ in range(2)]
    want = O.padding_masks(2, 3, metas, hw)
    got = torch.from_numpy(padding_mask_closed_form(metas, 3, hw)).bool()
    assert torch.equal(got, want)


def test_head_state_dict_contract(lib_path, golden_dir):
    """same keys/shapes as the reference head; six aliased branch slots share storage; flat layout is total."""
    import petr_amd
    from oracle import petr_oracle as O
    torch.manual_seed(0)
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900))
    head.init_weights()
    want = {l.split(' ')[0]: l.split(' ', 1)[1].strip() for l in open(os.path.join(golden_dir, 'state_dict_keys_petr.txt'))}
    got = {k: str(tuple(v.shape)) for k, v in head.state_dict().items()}
    assert got == want
    # seeded init reproduces the reference's draws (same construction order)
    ref = O.seeded_head(0, None, num_query=900).state_dict()
    assert all(torch.equal(head.state_dict()[k], ref[k]) for k in ref)
    head._ensure_flat()
    assert head.cls_branches[0][0].weight.data_ptr() == head.cls_branches[5][0].weight.data_ptr()
    assert head.reg_branches[0][4].bias.data_ptr() == head.reg_branches[3][4].bias.data_ptr()
    nparam = sum(p.numel() for p in head.parameters())
    assert nparam == 11091872 + 10                     # SURVEY §8(e) + code_weights
    assert nparam <= head._flat.numel() <= nparam + 4 * 222
    # every parameter is a view of the flat buffer; loading a state dict writes through
    lo, hi = head._flat.data_ptr(), head._flat.data_ptr() + head._flat.numel() * 4
    assert all(lo <= p.data_ptr() < hi for p in head.parameters())
    other = O.seeded_head(2, 1234, num_query=900).state_dict()
    head.load_state_dict(other)
    assert all(lo <= p.data_ptr() < hi for p in head.parameters())
    assert all(torch.equal(head.state_dict()[k], other[k]) for k in other)
    # backward-stage buckets tile the trainable part of the buffer, in order
    b = head.gradient_buckets()
    assert b[0][0] == 0 and all(b[i][1] == b[i + 1][0] for i in range(len(b) - 1)) and len(b) == 8
    assert head._flat.numel() - b[-1][1] <= 16        # only code_weights (no grad) after the last bucket


def test_headv2_state_dict_contract(lib_path, golden_dir):
    """PETRv2Head: 334 reference keys (deep-copied branches, RegLayer, fpe), same seeded init as the reference."""
    import petr_amd
    from oracle import petr_oracle as O
    torch.manual_seed(3)
    head = petr_amd.build_head(petr_amd.petrv2_head_cfg(num_query=16))
    head.init_weights()
    want = {l.split(' ')[0]: l.split(' ', 1)[1].strip() for l in open(os.path.join(golden_dir, 'state_dict_keys_petrv2.txt'))}
    assert {k: str(tuple(v.shape)) for k, v in head.state_dict().items()} == want
    ref = O.seeded_head(3, None, num_query=16, v2=True, with_fpe=True, with_time=True, with_multi=True,
                        code_weights=[1.0] * 10).state_dict()
    assert all(torch.equal(head.state_dict()[k], ref[k]) for k in ref)
    head._ensure_flat()
    assert head.cls_branches[0][0].weight.data_ptr() != head.cls_branches[5][0].weight.data_ptr()   # deep copies
    lo, hi = head._flat.data_ptr(), head._flat.data_ptr() + head._flat.numel() * 4
    assert all(lo <= p.data_ptr() < hi for p in head.parameters())
    assert len(head.gradient_buckets()) == 8


def test_legacy_key_remap(lib_path):
    """reference petr_head.py:345-359: DETR-era names load into attentions.N / post_norm."""
    import petr_amd
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=16))
    sd = {k: v.clone() for k, v in head.state_dict().items()}
    legacy = {}
    for k, v in sd.items():
        k2 = k.replace('.attentions.0.', '.self_attn.').replace('.attentions.1.', '.multihead_attn.')
        k2 = k2.replace('.decoder.post_norm.', '.decoder.norm.')
        legacy[k2] = v + 1.0
    assert any('.self_attn.' in k for k in legacy)
    head.load_state_dict(legacy)      # no metadata -> version None -> remap
    assert torch.equal(head.transformer.decoder.post_norm.weight, sd['transformer.decoder.post_norm.weight'] + 1.0)
    assert torch.equal(head.transformer.decoder.layers[3].attentions[1].attn.in_proj_bias,
                       sd['transformer.decoder.layers.3.attentions.1.attn.in_proj_bias'] + 1.0)


def test_registry_builds_reference_config_names(lib_path):
    import petr_amd
    assert petr_amd.HEADS.get('PETRHead') is petr_amd.PETRHead
    assert petr_amd.TRANSFORMER.get('PETRTransformer') is petr_amd.PETRTransformer
    assert petr_amd.ATTENTION.get('PETRMultiheadAttention') is petr_amd.PETRMultiheadAttention
    assert petr_amd.TRANSFORMER_LAYER.get('PETRTransformerDecoderLayer') is petr_amd.PETRTransformerDecoderLayer
    assert petr_amd.TRANSFORMER_LAYER_SEQUENCE.get('PETRTransformerDecoder') is petr_amd.PETRTransformerDecoder
    assert petr_amd.POSITIONAL_ENCODING.get('SinePositionalEncoding3D') is petr_amd.SinePositionalEncoding3D
    pe = petr_amd.build_positional_encoding(dict(type='SinePositionalEncoding3D', num_feats=128, normalize=True))
    assert 'num_feats=128' in repr(pe)
    with pytest.raises(AssertionError):
        petr_amd.build_head(petr_amd.petr_head_cfg(positional_encoding=dict(type='SinePositionalEncoding3D', num_feats=64,
                                                                             normalize=True)))


def test_executor_layout_and_workspace_queries(lib_path):
    """petr_head_layout / workspace_bytes / stage ranges are pure host functions."""
    import ctypes as C
    import petr_amd
    from petr_amd import _C
    head = petr_amd.build_head(petr_amd.petr_head_cfg(num_query=900))
    cfg = head._base_config()
    cfg.B, cfg.N, cfg.H, cfg.W = 1, 6, 16, 44
    L = _C.lib()
    nbytes = L.petr_head_workspace_bytes(C.byref(cfg))
    assert 200e6 < nbytes < 1.5e9
    off, n = C.c_long(), C.c_long()
    assert L.petr_head_ws_view(C.byref(cfg), b'memory', C.byref(off), C.byref(n)) == 0 and n.value == 4224 * 256
    assert L.petr_head_ws_view(C.byref(cfg), b'attn_cross.5', C.byref(off), C.byref(n)) == 0 and n.value == 900 * 256
    assert L.petr_head_ws_view(C.byref(cfg), b'no_such_buffer', C.byref(off), C.byref(n)) != 0
    assert b'no_such_buffer' in L.petr_last_error()
    cfg.embed_dims = 128
    assert L.petr_head_workspace_bytes(C.byref(cfg)) == 0      # unsupported configs are refused, not guessed
    assert L.petr_mha_choose_split(1, 8, 900, 4224) >= 2
