"""ctypes binding of libpetr_hip.so (include/petr_hip.h).

The product path has NO fallback: if the shared library is missing or a call fails, an exception
is raised.  Nothing under ``oracle/`` is ever imported from here.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PETR_HIP_LIB: another build of the same ABI (A/B timing of kernel variants on ONE box; devices differ by several %)
LIB_PATH = os.environ.get('PETR_HIP_LIB') or os.path.join(_HERE, 'lib', 'libpetr_hip.so')

c_float_p = C.POINTER(C.c_float)


class PetrHipError(RuntimeError):
    pass


def _fields(*spec):
    return [(n, t) for n, t in spec]


class Coords3dArgs(C.Structure):
    _fields_ = _fields(('img2lidar', C.c_void_p), ('depth', C.c_void_p), ('out', C.c_void_p), ('cmask', C.c_void_p),
                       ('B', C.c_int), ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('D', C.c_int),
                       ('pad_h', C.c_float), ('pad_w', C.c_float), ('range', C.c_float * 6), ('eps', C.c_float))


class Sine3dArgs(C.Structure):
    _fields_ = _fields(('mask', C.c_void_p), ('dim_t', C.c_void_p), ('out', C.c_void_p),
                       ('B', C.c_int), ('N', C.c_int), ('H', C.c_int), ('W', C.c_int), ('F', C.c_int),
                       ('normalize', C.c_int), ('scale', C.c_float), ('eps', C.c_float), ('offset', C.c_float))


class Dropout(C.Structure):          # petr_dropout
    _fields_ = _fields(('seed', C.c_uint64), ('site', C.c_uint32), ('p', C.c_float))


def dropout(spec):
    """(seed, site, p) or None -> petr_dropout (p = 0: off)."""
    if spec is None:
        return Dropout(0, 0, 0.0)
    seed, site, p = spec
    return Dropout(int(seed) & 0xFFFFFFFFFFFFFFFF, int(site), float(p))


class GemmArgs(C.Structure):
    _fields_ = _fields(
        ('a', C.c_void_p), ('lda', C.c_long), ('a_kcontig', C.c_int), ('a_bs0', C.c_long), ('a_bs1', C.c_long),
        ('a2', C.c_void_p), ('a2_rows', C.c_int), ('a2_ncols', C.c_int),
        ('b', C.c_void_p), ('ldb', C.c_long), ('b_kcontig', C.c_int), ('b_bs0', C.c_long), ('b_bs1', C.c_long),
        ('c', C.c_void_p), ('ldc', C.c_long), ('c_bs0', C.c_long), ('c_bs1', C.c_long), ('c_nblk', C.c_int),
        ('c_nblk_stride', C.c_long),
        ('bias', C.c_void_p), ('bias_bs0', C.c_long), ('bias_bs1', C.c_long),
        ('r', C.c_void_p), ('ldr', C.c_long), ('r_bs0', C.c_long), ('r_bs1', C.c_long),
        ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('nb0', C.c_int), ('nb1', C.c_int),
        ('split_k', C.c_int), ('c_split_stride', C.c_long), ('k_seg', C.c_int), ('a_seg_stride', C.c_long),
        ('b_seg_stride', C.c_long), ('a_colsum', C.c_void_p), ('cs_bs0', C.c_long), ('cs_bs1', C.c_long),
        ('flags', C.c_int), ('alpha', C.c_float), ('drop', Dropout))


GEMM_RELU, GEMM_ACCUMULATE, GEMM_RELU_MASK, GEMM_SIGMOID_MUL, GEMM_ATOMIC, GEMM_STORE_BF16, GEMM_BF16 = 1, 2, 4, 8, 16, 32, 64
GEMM_BIAS_M, GEMM_A_BF16, GEMM_B_BF16, GEMM_R_BF16 = 128, 256, 512, 1024
LN_RELU, LN_NAN_TO_NUM = 1, 2


class LayerNormArgs(C.Structure):
    _fields_ = _fields(('x', C.c_void_p), ('n_partials', C.c_int), ('partial_stride', C.c_long),
                       ('bias', C.c_void_p), ('residual', C.c_void_p), ('gamma', C.c_void_p), ('beta', C.c_void_p),
                       ('y', C.c_void_p), ('z_out', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p),
                       ('M', C.c_int), ('C', C.c_int), ('eps', C.c_float), ('flags', C.c_int),
                       ('y2', C.c_void_p), ('add2', C.c_void_p), ('add2_rows', C.c_int), ('drop', Dropout))


class LayerNormBwdArgs(C.Structure):
    _fields_ = _fields(('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('gamma', C.c_void_p),
                       ('dy', C.c_void_p), ('y', C.c_void_p), ('dz', C.c_void_p), ('dgamma', C.c_void_p),
                       ('dbeta', C.c_void_p), ('ws', C.c_void_p), ('M', C.c_int), ('C', C.c_int), ('flags', C.c_int),
                       ('dz_accumulate', C.c_int), ('dy_partials', C.c_int), ('dy_partial_stride', C.c_long),
                       ('dy_residual', C.c_void_p), ('dz_drop', C.c_void_p), ('drop', Dropout))


class MhaFwdArgs(C.Structure):
    _fields_ = _fields(
        ('q', C.c_void_p), ('q_bs', C.c_long), ('q_hs', C.c_long), ('q_rs', C.c_long),
        ('k', C.c_void_p), ('k_bs', C.c_long), ('k_hs', C.c_long), ('k_rs', C.c_long),
        ('v', C.c_void_p), ('v_bs', C.c_long), ('v_hs', C.c_long), ('v_rs', C.c_long),
        ('o', C.c_void_p), ('o_bs', C.c_long), ('o_hs', C.c_long), ('o_rs', C.c_long),
        ('lse', C.c_void_p), ('kpm', C.c_void_p),
        ('B', C.c_int), ('H', C.c_int), ('Q', C.c_int), ('L', C.c_int), ('scale', C.c_float),
        ('n_split', C.c_int), ('ws', C.c_void_p), ('ws_bytes', C.c_size_t), ('drop', Dropout), ('sched', C.c_void_p),
        ('drop_bits', C.c_void_p), ('defer_merge', C.c_int))


class AttnOutLnArgs(C.Structure):
    _fields_ = _fields(
        ('a', C.c_void_p), ('o_part', C.c_void_p), ('ml_part', C.c_void_p), ('n_split', C.c_int),
        ('B', C.c_int), ('H', C.c_int), ('Q', C.c_int), ('attn_scale', C.c_float), ('lse', C.c_void_p),
        ('wT', C.c_void_p), ('bias', C.c_void_p), ('residual', C.c_void_p), ('drop', Dropout),
        ('gamma', C.c_void_p), ('beta', C.c_void_p), ('eps', C.c_float),
        ('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('y', C.c_void_p),
        ('y2', C.c_void_p), ('add2', C.c_void_p), ('add2_rows', C.c_int), ('M', C.c_int),
        ('w2T', C.c_void_p), ('bias2', C.c_void_p), ('out2', C.c_void_p))


class LnProjArgs(C.Structure):
    _fields_ = _fields(
        ('x', C.c_void_p), ('n_partials', C.c_int), ('partial_stride', C.c_long), ('bias', C.c_void_p), ('residual', C.c_void_p),
        ('drop', Dropout), ('gamma', C.c_void_p), ('beta', C.c_void_p), ('eps', C.c_float),
        ('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('y', C.c_void_p),
        ('y2', C.c_void_p), ('add2', C.c_void_p), ('add2_rows', C.c_int), ('M', C.c_int),
        ('w2T', C.c_void_p), ('bias2', C.c_void_p), ('out2', C.c_void_p), ('n2', C.c_int), ('n2_pos', C.c_int),
        ('out2_bf16', C.c_void_p))


class LnBwdProjArgs(C.Structure):
    _fields_ = _fields(
        ('z', C.c_void_p), ('mean', C.c_void_p), ('rstd', C.c_void_p), ('gamma', C.c_void_p),
        ('dy', C.c_void_p), ('dy_partials', C.c_int), ('dy_partial_stride', C.c_long), ('dy_residual', C.c_void_p),
        ('dz', C.c_void_p), ('dz_drop', C.c_void_p), ('drop', Dropout), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p),
        ('M', C.c_int), ('w', C.c_void_p), ('n2', C.c_int), ('alpha', C.c_float), ('relu_mask', C.c_void_p), ('out', C.c_void_p),
        ('pre_a', C.c_void_p), ('pre_w', C.c_void_p), ('pre_n', C.c_int))


class FfnFwdArgs(C.Structure):
    _fields_ = _fields(
        ('x', C.c_void_p), ('w1t', C.c_void_p), ('b1', C.c_void_p), ('w2t', C.c_void_p),
        ('hidden', C.c_void_p), ('part', C.c_void_p), ('part_stride', C.c_long), ('drop', Dropout),
        ('M', C.c_int), ('F', C.c_int), ('n_split', C.c_int))


class FfnBwdArgs(C.Structure):
    _fields_ = _fields(
        ('dy', C.c_void_p), ('w2', C.c_void_p), ('hidden', C.c_void_p), ('alpha', C.c_float), ('w1', C.c_void_p),
        ('d_hidden', C.c_void_p), ('part', C.c_void_p), ('part_stride', C.c_long),
        ('M', C.c_int), ('F', C.c_int), ('n_split', C.c_int))


class WgradItem(C.Structure):        # petr_wgrad_item
    _fields_ = _fields(('dy', C.c_void_p), ('lda', C.c_long), ('x', C.c_void_p), ('ldb', C.c_long), ('dw', C.c_void_p),
                       ('ldc', C.c_long), ('db', C.c_void_p), ('M', C.c_int), ('N', C.c_int), ('K', C.c_int), ('ksplit', C.c_int))


class BranchFwdArgs(C.Structure):    # petr_branch_fwd_args
    _fields_ = _fields(('x', C.c_void_p), ('w1t', C.c_void_p), ('b1', C.c_void_p), ('g1', C.c_void_p), ('be1', C.c_void_p),
                       ('w2t', C.c_void_p), ('b2', C.c_void_p), ('g2', C.c_void_p), ('be2', C.c_void_p),
                       ('w3', C.c_void_p), ('b3', C.c_void_p), ('param_gs', C.c_long), ('wt_gs', C.c_long),
                       ('h1', C.c_void_p), ('y1', C.c_void_p), ('h2', C.c_void_p), ('y2', C.c_void_p),
                       ('mean1', C.c_void_p), ('rstd1', C.c_void_p), ('mean2', C.c_void_p), ('rstd2', C.c_void_p),
                       ('out', C.c_void_p), ('n_out', C.c_int), ('rows', C.c_int), ('groups', C.c_int), ('eps', C.c_float))


class BranchBwdArgs(C.Structure):    # petr_branch_bwd_args
    _fields_ = _fields(('d_out', C.c_void_p), ('n_out', C.c_int), ('w3', C.c_void_p), ('d_y2', C.c_void_p),
                       ('y2', C.c_void_p), ('h2', C.c_void_p), ('mean2', C.c_void_p), ('rstd2', C.c_void_p), ('g2', C.c_void_p),
                       ('w2', C.c_void_p),
                       ('y1', C.c_void_p), ('h1', C.c_void_p), ('mean1', C.c_void_p), ('rstd1', C.c_void_p), ('g1', C.c_void_p),
                       ('w1', C.c_void_p), ('param_gs', C.c_long),
                       ('d_h2', C.c_void_p), ('d_h1', C.c_void_p), ('d_x', C.c_void_p),
                       ('dg2', C.c_void_p), ('dbe2', C.c_void_p), ('dg1', C.c_void_p), ('dbe1', C.c_void_p),
                       ('rows', C.c_int), ('groups', C.c_int), ('dw3', C.c_void_p), ('db3', C.c_void_p))


class TaskHeadsFwdArgs(C.Structure):    # petr_task_heads_fwd_args
    _fields_ = _fields(('h', C.c_void_p), ('w2', C.c_void_p), ('b2', C.c_void_p), ('param_gs', C.c_long), ('head_stride', C.c_long),
                       ('out', C.c_void_p), ('ld_out', C.c_int), ('rows', C.c_int), ('groups', C.c_int), ('heads', C.c_int),
                       ('dims', C.c_int * 8), ('cols', C.c_int * 8))


class TaskHeadsBwdArgs(C.Structure):    # petr_task_heads_bwd_args
    _fields_ = _fields(('d_out', C.c_void_p), ('ld_out', C.c_int), ('h', C.c_void_p), ('w2', C.c_void_p), ('param_gs', C.c_long),
                       ('head_stride', C.c_long), ('d_h', C.c_void_p), ('dw2', C.c_void_p), ('db2', C.c_void_p),
                       ('rows', C.c_int), ('groups', C.c_int), ('heads', C.c_int), ('dims', C.c_int * 8), ('cols', C.c_int * 8))


class MhaBwdArgs(C.Structure):
    _fields_ = _fields(
        ('q', C.c_void_p), ('q_bs', C.c_long), ('q_hs', C.c_long), ('q_rs', C.c_long),
        ('k', C.c_void_p), ('k_bs', C.c_long), ('k_hs', C.c_long), ('k_rs', C.c_long),
        ('v', C.c_void_p), ('v_bs', C.c_long), ('v_hs', C.c_long), ('v_rs', C.c_long),
        ('o', C.c_void_p), ('o_bs', C.c_long), ('o_hs', C.c_long), ('o_rs', C.c_long),
        ('d_o', C.c_void_p), ('do_bs', C.c_long), ('do_hs', C.c_long), ('do_rs', C.c_long),
        ('lse', C.c_void_p), ('kpm', C.c_void_p),
        ('dq', C.c_void_p), ('dq_bs', C.c_long), ('dq_hs', C.c_long), ('dq_rs', C.c_long),
        ('dk', C.c_void_p), ('dk_bs', C.c_long), ('dk_hs', C.c_long), ('dk_rs', C.c_long),
        ('dv', C.c_void_p), ('dv_bs', C.c_long), ('dv_hs', C.c_long), ('dv_rs', C.c_long),
        ('B', C.c_int), ('H', C.c_int), ('Q', C.c_int), ('L', C.c_int), ('scale', C.c_float),
        ('ws', C.c_void_p), ('ws_bytes', C.c_size_t), ('drop', Dropout), ('drop_bits', C.c_void_p))


class MhaBwdBf16Args(C.Structure):    # petr_mha_bwd_bf16_args = petr_mha_bwd_args (k / v bf16) + dkv_overwrite, dkv_bf16
    _fields_ = list(MhaBwdArgs._fields_) + [('dkv_overwrite', C.c_int), ('dkv_bf16', C.c_int)]


class BboxArgs(C.Structure):
    _fields_ = _fields(('reg', C.c_void_p), ('ref', C.c_void_p), ('out', C.c_void_p),
                       ('rows', C.c_int), ('Q', C.c_int), ('code', C.c_int),
                       ('pc_range', C.c_float * 6), ('time_div', C.c_float), ('eps', C.c_float))


class LossArgs(C.Structure):         # petr_loss_args
    _fields_ = _fields(('cls', C.c_void_p), ('box', C.c_void_p), ('gt_boxes', C.c_void_p), ('gt_labels', C.c_void_p),
                       ('gt_offsets', C.c_void_p),
                       ('NL', C.c_int), ('B', C.c_int), ('Q', C.c_int), ('NC', C.c_int), ('CS', C.c_int),
                       ('Gtot', C.c_int), ('Gmax', C.c_int), ('num_pos', C.c_long),
                       ('cls_weight', C.c_float), ('bbox_weight', C.c_float), ('alpha', C.c_float), ('gamma', C.c_float),
                       ('bg_cls_weight', C.c_float), ('code_weights', C.c_float * 10),
                       ('losses', C.c_void_p), ('d_cls', C.c_void_p), ('d_box', C.c_void_p), ('assigned', C.c_void_p),
                       ('ws', C.c_void_p), ('ws_bytes', C.c_size_t), ('avg_factors', C.c_void_p))


class DecodeArgs(C.Structure):       # petr_decode_args
    _fields_ = _fields(('bbox_preds', C.c_void_p), ('index', C.c_void_p), ('scores', C.c_void_p), ('boxes', C.c_void_p),
                       ('labels', C.c_void_p), ('keep', C.c_void_p), ('n', C.c_int), ('num_classes', C.c_int),
                       ('code', C.c_int), ('post_center_range', C.c_float * 6), ('score_threshold', C.c_float),
                       ('bottom_center', C.c_int))


class DecodeTopkArgs(C.Structure):   # petr_decode_topk_args
    _fields_ = _fields(('cls_scores', C.c_void_p), ('bbox_preds', C.c_void_p), ('boxes', C.c_void_p), ('scores', C.c_void_p),
                       ('labels', C.c_void_p), ('keep', C.c_void_p), ('index', C.c_void_p),
                       ('B', C.c_int), ('Q', C.c_int), ('num_classes', C.c_int), ('code', C.c_int), ('k', C.c_int),
                       ('post_center_range', C.c_float * 6), ('score_threshold', C.c_float), ('bottom_center', C.c_int))


class HeadConfig(C.Structure):
    _fields_ = _fields(
        ('B', C.c_int), ('N', C.c_int), ('C_in', C.c_int), ('H', C.c_int), ('W', C.c_int),
        ('num_query', C.c_int), ('num_layers', C.c_int), ('num_heads', C.c_int), ('embed_dims', C.c_int),
        ('ffn_dims', C.c_int), ('depth_num', C.c_int), ('num_classes', C.c_int), ('code_size', C.c_int),
        ('v2', C.c_int), ('with_fpe', C.c_int), ('with_time', C.c_int), ('with_multi', C.c_int),
        ('shared_branches', C.c_int), ('LID', C.c_int), ('depth_start', C.c_float),
        ('position_range', C.c_float * 6), ('pc_range', C.c_float * 6), ('pad_h', C.c_float), ('pad_w', C.c_float),
        ('has_mask', C.c_int), ('training', C.c_int))


MAX_PARAMS = 512


class HeadLayout(C.Structure):
    _fields_ = _fields(('count', C.c_int), ('total', C.c_long), ('name', (C.c_char * 96) * MAX_PARAMS),
                       ('offset', C.c_long * MAX_PARAMS), ('ndim', C.c_int * MAX_PARAMS),
                       ('shape', (C.c_int * 4) * MAX_PARAMS), ('alias_of', C.c_int * MAX_PARAMS))


class HeadIO(C.Structure):
    _fields_ = _fields(('params', C.c_void_p), ('feats', C.c_void_p), ('img2lidar', C.c_void_p), ('depth', C.c_void_p),
                       ('dim_t', C.c_void_p), ('mask', C.c_void_p), ('time_div', C.c_float),
                       ('all_cls_scores', C.c_void_p), ('all_bbox_preds', C.c_void_p),
                       ('ws', C.c_void_p), ('ws_bytes', C.c_size_t), ('ctx', C.c_void_p),
                       ('dropout_p', C.c_float), ('dropout_seed', C.c_uint64), ('attn_bf16', C.c_int),
                       ('memory_in', C.c_void_p))


class HeadGrads(C.Structure):
    _fields_ = _fields(('d_cls', C.c_void_p), ('d_bbox', C.c_void_p), ('d_params', C.c_void_p), ('d_feats', C.c_void_p))


_lib = None


def lib():
    """Load libpetr_hip.so (once).  Raises PetrHipError if it is not built — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PetrHipError(
            f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            f'or `make -C petr_amd/csrc`. petr_amd has no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    L.petr_version.restype = C.c_int
    L.petr_last_error.restype = C.c_char_p
    L.petr_device_caps.argtypes = [C.POINTER(C.c_int), C.c_char_p, C.c_int]
    L.petr_coords3d_fwd.argtypes = [C.POINTER(Coords3dArgs), C.c_void_p]
    L.petr_sine3d_fwd.argtypes = [C.POINTER(Sine3dArgs), C.c_void_p]
    L.petr_posemb3d_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.petr_posemb3d_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.petr_gemm.argtypes = [C.POINTER(GemmArgs), C.c_void_p]
    L.petr_colsum_workspace_bytes.argtypes = [C.c_int]
    L.petr_colsum_workspace_bytes.restype = C.c_size_t
    L.petr_colsum.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.petr_layernorm_fwd.argtypes = [C.POINTER(LayerNormArgs), C.c_void_p]
    L.petr_layernorm_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.petr_layernorm_bwd_workspace_bytes.restype = C.c_size_t
    L.petr_layernorm_bwd.argtypes = [C.POINTER(LayerNormBwdArgs), C.c_void_p]
    L.petr_mha_fwd_workspace_bytes.argtypes = [C.c_int] * 5
    L.petr_mha_fwd_workspace_bytes.restype = C.c_size_t
    L.petr_mha_choose_split.argtypes = [C.c_int] * 4
    L.petr_mha_fwd.argtypes = [C.POINTER(MhaFwdArgs), C.c_void_p]
    L.petr_mha_fwd_bf16_choose_split.argtypes = [C.c_int] * 4
    L.petr_mha_fwd_bf16_workspace_bytes.argtypes = [C.c_int] * 5
    L.petr_mha_fwd_bf16_workspace_bytes.restype = C.c_size_t
    L.petr_mha_fwd_bf16.argtypes = [C.POINTER(MhaFwdArgs), C.c_void_p]   # same block, k / v are bf16
    L.petr_cast_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.petr_dropout_bits_words.argtypes = [C.c_int] * 3
    L.petr_dropout_bits_words.restype = C.c_size_t
    L.petr_dropout_bits.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.petr_mha_bwd_workspace_bytes.argtypes = [C.c_int] * 4
    L.petr_mha_bwd_workspace_bytes.restype = C.c_size_t
    L.petr_mha_bwd.argtypes = [C.POINTER(MhaBwdArgs), C.c_void_p]
    L.petr_mha_bwd_bf16_workspace_bytes.argtypes = [C.c_int] * 4
    L.petr_mha_bwd_bf16_workspace_bytes.restype = C.c_size_t
    L.petr_mha_bwd_bf16.argtypes = [C.POINTER(MhaBwdBf16Args), C.c_void_p]
    L.petr_bbox_epilogue_fwd.argtypes = [C.POINTER(BboxArgs), C.c_void_p]
    L.petr_bbox_epilogue_bwd.argtypes = [C.POINTER(BboxArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.petr_prof_begin.argtypes = [C.c_int]
    L.petr_prof_end.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]
    L.petr_add_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
    L.petr_add_rows_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
    L.petr_gate_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.petr_gate_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_void_p]
    L.petr_fill.argtypes = [C.c_void_p, C.c_float, C.c_long, C.c_void_p]
    L.petr_fpn_upsample_add_bwd.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long] + [C.c_int] * 7 + [C.c_void_p]
    L.petr_nchw_to_padded_nhwc.argtypes = [C.c_void_p, C.c_void_p] + [C.c_int] * 4 + [C.c_void_p]
    L.petr_add_rows2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    L.petr_add_rows2_bf16.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
    L.petr_wgrad_grouped.argtypes = [C.POINTER(WgradItem), C.c_int, C.c_void_p]
    L.petr_branch_fwd.argtypes = [C.POINTER(BranchFwdArgs), C.c_void_p]
    L.petr_branch_bwd.argtypes = [C.POINTER(BranchBwdArgs), C.c_void_p]
    L.petr_task_heads_fwd.argtypes = [C.POINTER(TaskHeadsFwdArgs), C.c_void_p]
    L.petr_task_heads_bwd.argtypes = [C.POINTER(TaskHeadsBwdArgs), C.c_void_p]
    L.petr_fpn_upsample_add.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_long, C.c_long, C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
    L.petr_axpy.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_long, C.c_void_p]
    L.petr_reduce_partials.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long,
                                       C.c_int, C.c_void_p]
    L.petr_reduce_batch.argtypes = [C.c_void_p, C.c_int, C.c_long, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:
        raise PetrHipError(f'{LIB_PATH} is stale: missing exports {missing}; rebuild it')
    L.petr_ctx_create.argtypes = [C.POINTER(C.c_void_p), C.c_int]
    L.petr_ctx_destroy.argtypes = [C.c_void_p]
    L.petr_ctx_side_stream.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.petr_ctx_join_into.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.petr_head_layout.argtypes = [C.POINTER(HeadConfig), C.POINTER(HeadLayout)]
    L.petr_head_workspace_bytes.argtypes = [C.POINTER(HeadConfig)]
    L.petr_head_workspace_bytes.restype = C.c_size_t
    L.petr_head_fwd.argtypes = [C.POINTER(HeadConfig), C.POINTER(HeadIO), C.c_void_p]
    L.petr_head_bwd_num_stages.argtypes = [C.POINTER(HeadConfig)]
    L.petr_head_bwd_stage_range.argtypes = [C.POINTER(HeadConfig), C.c_int, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.petr_head_bwd.argtypes = [C.POINTER(HeadConfig), C.POINTER(HeadIO), C.POINTER(HeadGrads), C.c_int, C.c_int,
                                C.c_void_p]
    L.petr_loss_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.petr_loss_workspace_bytes.restype = C.c_size_t
    L.petr_head_ws_view.argtypes = [C.POINTER(HeadConfig), C.c_char_p, C.POINTER(C.c_long), C.POINTER(C.c_long)]
    _lib = L
    return L


EXPORTS = [
    'petr_version', 'petr_last_error', 'petr_device_caps', 'petr_coords3d_fwd', 'petr_sine3d_fwd',
    'petr_posemb3d_fwd', 'petr_posemb3d_bwd', 'petr_gemm', 'petr_colsum_workspace_bytes', 'petr_colsum',
    'petr_layernorm_fwd', 'petr_layernorm_bwd_workspace_bytes', 'petr_layernorm_bwd',
    'petr_mha_fwd_workspace_bytes', 'petr_mha_choose_split', 'petr_mha_fwd', 'petr_mha_fwd_bf16_workspace_bytes',
    'petr_mha_fwd_bf16', 'petr_mha_fwd_bf16_choose_split', 'petr_attn_out_ln', 'petr_ln_proj', 'petr_ln_bwd_proj', 'petr_ffn_fwd', 'petr_ffn_bwd', 'petr_cast_bf16', 'petr_add_rows_bf16', 'petr_mha_bwd_workspace_bytes',
    'petr_mha_bwd', 'petr_mha_bwd_bf16_workspace_bytes', 'petr_mha_bwd_bf16', 'petr_bbox_epilogue_fwd', 'petr_bbox_epilogue_bwd', 'petr_fill', 'petr_axpy', 'petr_add_rows', 'petr_gate_fwd', 'petr_gate_bwd', 'petr_prof_begin', 'petr_prof_end',
    'petr_reduce_partials', 'petr_reduce_batch', 'petr_head_layout', 'petr_head_workspace_bytes', 'petr_head_fwd',
    'petr_head_bwd_num_stages', 'petr_head_bwd_stage_range', 'petr_head_bwd', 'petr_head_ws_view',
    'petr_ctx_create', 'petr_ctx_destroy', 'petr_ctx_join_into', 'petr_ctx_side_stream', 'petr_dropout_mask', 'petr_dropout_bits_words', 'petr_dropout_bits', 'petr_loss_workspace_bytes', 'petr_loss_fwd_bwd',
    'petr_decode_boxes', 'petr_decode_topk', 'petr_fpn_upsample_add', 'petr_wgrad_grouped', 'petr_branch_fwd', 'petr_branch_bwd', 'petr_task_heads_fwd', 'petr_task_heads_bwd', 'petr_fpn_upsample_add_bwd', 'petr_nchw_to_padded_nhwc', 'petr_add_rows2', 'petr_add_rows2_bf16',
]


def check(status, what='libpetr_hip'):
    if status != 0:
        msg = lib().petr_last_error()
        raise PetrHipError(f'{what} failed ({status}): {msg.decode() if msg else "?"}')
