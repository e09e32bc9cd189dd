"""ctypes binding of libtarl_hip.so (C ABI: include/tarl_hip.h).

There is NO fallback: if the shared library is missing or lacks a symbol, ``load()`` raises. Build it with
``python __graft_entry__.py build`` (or ``make -C tarl-simulator_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# TARL_HIP_LIB (developer knob): another build of the same library, e.g. an A/B variant under tmp_ab/ (tools/ab.sh)
LIB_PATH = os.environ.get("TARL_HIP_LIB") or os.path.join(_HERE, "libtarl_hip.so")

_p = C.c_void_p
_i64 = C.c_int64
_i32 = C.c_int32
_u64 = C.c_uint64
_f32 = C.c_float
_f64 = C.c_double

# name -> (restype, argtypes); must list every symbol include/tarl_hip.h declares (tests/test_abi.py checks it)
_STATE = [_p, _i64, _i64, _i64, _i32]  # x, B, x_bstride, ldx, Nmax
SIGNATURES = {
    "tarl_abi_version": (C.c_int, []),
    "tarl_build_flags": (C.c_char_p, []),
    "tarl_last_error": (C.c_char_p, []),
    "tarl_plan_create": (C.c_int, [_p, _i64, _i64, _p, C.POINTER(_p)]),
    "tarl_plan_destroy": (None, [_p]),
    "tarl_plan_info": (C.c_int, [_p, C.POINTER(_i64)]),
    "tarl_plan_geometry": (C.c_int, [_p, C.POINTER(_i64)]),
    "tarl_direction_step": (C.c_int, [_p] + _STATE + [_i64, _p, _p, _f32, _p, _f32, _p, _u64, _u64, _p, _p, _p, _p]),
    "tarl_response_step": (C.c_int, [_p] + _STATE + [_i64, _p, _p, _p]),
    "tarl_core_step": (C.c_int, [_p] + _STATE + [_i64, _p, _p, _f32, _p, _f32, _p, _u64, _u64, _p, _p, _p, _p, _p, _p]),
    "tarl_apply_action": (C.c_int, [_p] + _STATE + [_p, _p, _p]),
    "tarl_withdraw_step": (C.c_int, [_p] + _STATE + [_i64, _p, _i64, _i64, _f32, _p, _p]),
    "tarl_insert_step": (C.c_int, _STATE + [_i64, _p, _i64, _i64, _p, _f32, _p, _p, _p, _p]),
    "tarl_reset_state": (C.c_int, _STATE + [_i64, _p, _i64, _i64, _p]),
    "tarl_graphdist_softmax": (C.c_int, [_p, _p, _i64, _f32, _p, _p]),
    "tarl_graphdist_sample": (C.c_int, [_p, _p, _i64, _p, _u64, _u64, _p, _p, _p, _p]),
    "tarl_graphdist_rollout_scratch_bytes": (C.c_int64, [_p, _i64]),
    "tarl_graphdist_rollout": (C.c_int, [_p, _p, _i64, _f32, _p, _u64, _u64, _p, _p, _p, _p, _p, _p]),
    "tarl_graphdist_mode": (C.c_int, [_p, _p, _i64, _p, _p, _p]),
    "tarl_graphdist_logprob_entropy_fwd": (C.c_int, [_p, _p, _i64, _p, _p, _p, _p, _p]),
    "tarl_graphdist_logprob_entropy_bwd": (C.c_int, [_p, _p, _i64, _f32, _p, _p, _p, _p, _p, _p, _p]),
    "tarl_policy_edge_logits_fwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _i64, _p, _p]),
    "tarl_policy_edge_logits_bwd_scratch_floats": (_i64, [_p, _i64]),
    "tarl_policy_edge_logits_bwd": (C.c_int, [_p, _p, _i64, _i64, _i64, _p, _p, _i64, _p, _p]),
    "tarl_policy_obs16": (C.c_int, [_p, _i64, _p, _p, _i64, _i64, _i64, _i64, _p, _p]),
    "tarl_fused_obs16": (C.c_int, [_p, _p] + _STATE + [_p, _i64, _i64, _p, _p]),
    "tarl_fused_obs16_bf16": (C.c_int, [_p, _p] + _STATE + [_p, _i64, _i64, _p, _p]),
    "tarl_fused_obs16_rows": (C.c_int, [_p, _p] + _STATE + [_p, _i64, _i64, _p, _p, _i64, _p, _p]),
    "tarl_policy_edge_mlp_fwd": (C.c_int, [_p, _p, _i64, _p] + [_p] * 6 + [C.c_int, _p, _p]),
    "tarl_policy_edge_mlp_bwd_scratch_floats": (_i64, [_p, _i64]),
    "tarl_policy_edge_mlp_bwd": (C.c_int, [_p, _p, _i64, _p] + [_p] * 6 + [_p, _p] + [_p] * 6 + [_p]),
    "tarl_critic_mlp_fwd": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 6 + [_p, _p, _p, _p]),
    "tarl_critic_splitk_scratch_floats": (C.c_int64, [_i64, _i64]),
    "tarl_critic_mlp_fwd_splitk": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 6 + [_p, _p, _p, _p, _p]),
    "tarl_critic_mlp_bwd_scratch_floats": (_i64, [_i64, _i64]),
    "tarl_critic_mlp_bwd": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 3 + [_p, _p, _p, _p] + [_p] * 6 + [_p]),
    "tarl_value_mpnn_fwd": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p]),
    "tarl_value_mpnn_bwd": (C.c_int, [_p, _p, _i64, _p, _p, _i64, _p, _p, _p, _p, _p, _p, _p]),
    "tarl_gae": (C.c_int, [_p, _p, _p, _p, _p, _i64, _i64, _f32, _f32, _p, _p, _p]),
    "tarl_advantage_stats": (C.c_int, [_p, _i64, _p, _p, _p]),
    "tarl_advantage_normalize": (C.c_int, [_p, _i64, _p, _p]),
    "tarl_ppo_loss": (C.c_int, [_p] * 6 + [_i64, _f32, _f32, _f32, _f32, _p, _p, _p, _p, _p]),
    "tarl_adam_step": (C.c_int, [_p, _p, _p, _p, _i64, _i64, _f64, _f64, _f64, _f64, _f32, _p]),
    "tarl_critic_mlp_fwd_slabs": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 6 + [_p, _p]),
    "tarl_critic_mlp_fwd_u8": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 6 + [_p, _p, _p, _p]),
    "tarl_critic_mlp_fwd_slabs_u8": (C.c_int, [_p, _i64, _i64, _i64, _p, _i64] + [_p] * 6 + [_p, _p, _p]),
    "tarl_critic_split_scratch_bytes": (_i64, [_i64]),
    "tarl_fused_slot_floats": (_i64, [_i32]),
    "tarl_fused_bufs_bytes": (_i64, []),
    "tarl_fused_pack": (C.c_int, [_p, _p] + _STATE + [_p, _p, _p, _i64, _i64, _p]),
    "tarl_fused_reset": (C.c_int, [_p, _p, _i64, _i32, _p, _i64, _i64, _p]),
    "tarl_fused_export": (C.c_int, [_p, _p] + _STATE + [_f32, _p]),
    "tarl_fused_policy_prepare": (C.c_int, [_p, _p, _p, _i64, _f32, _p, _p, _p, _p, _p]),
    "tarl_fused_frame": (C.c_int, [_p, _p, _i64, _i32, _p, _p, _p, _p, _u64, _u64, _p, _i64, _i64, _p, _p, _f32, C.c_int,
                                   _f32, _f32, _p, _u64, _u64] + [_p] * 10),
    "tarl_fused_apply_choice": (C.c_int, [_p, _p, _i64, _p, _p]),
    "tarl_fused_rollout": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _f32, _p, _p, _p, _u64, _u64, _p, _i64, _i64, _p, _p,
                                     _f32, C.c_int, _u64, _u64] + [_p] * 9 + [_i32, _p, _p, _p, _p]),
    "tarl_fused_rollout_scratch_ints": (_i64, [_p, _i64, _i64]),
    "tarl_fused_set_actions": (C.c_int, [_p, _p, _i64, _p, _p]),
    "tarl_fused_rollout_policy": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _f32, _p, _i64, _i64, _p, _i64, _i64, _p, _p,
                                            _f32, C.c_int] + [_p] * 6 + [C.c_int, _f32, _u64, _u64, _u64, _u64] +
                                  [_p] * 12 + [_i32, _p, _p, _p, _p]),
    "tarl_rollout_gather": (C.c_int, [_p, _p, _p, _i64, _i64, C.c_int, _p, _i64, _p, _p, _p]),
    "tarl_rollout_env_supported": (C.c_int, [_p]),
    "tarl_rollout_env_scratch_bytes": (_i64, [_p]),
    "tarl_rollout_env": (C.c_int, [_p, _p, _i64, _i32, _i64, _p, _f32, _p, _p, _p, _u64, _u64, _p, _i64, _i64, _p, _p,
                                   _f32, C.c_int, _u64, _u64] + [_p] * 7 + [_i32, _p, _p, _p, _p]),
    "tarl_edge_travel_time": (C.c_int, [_p] + _STATE + [_p, _p, _p]),
    "tarl_apsp_scratch_bytes": (_i64, [_p, _i64]),
    "tarl_apsp": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p, _p]),
    "tarl_apsp_f64": (C.c_int, [_p, _p, _i64, _i64, _p, _i64, _p, _p, _p]),
    "tarl_msa_assign": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _p, _p, _p]),
    "tarl_select_next_hop": (C.c_int, _STATE + [_i64, _p, _i64, _i64, _p, _i64, _p]),
    "tarl_noise_export": (C.c_int, [_p, C.c_int, _u64, _u64, _p, _i64, _p, _p]),
    "tarl_prof_enable": (C.c_int, [_i64]),
    "tarl_prof_collect": (C.c_int, [_i64, C.POINTER(_f64), C.POINTER(_f64), C.POINTER(_i64)]),
}



class FusedStruct(C.Structure):
    """``tarl_fused`` of include/tarl_hip.h."""
    _fields_ = ([(n, C.c_void_p) for n in ("hdp", "tl", "gc8", "post", "st0", "slots")] +
                [("ld_slots", C.c_int64)] +
                [(n, C.c_void_p) for n in ("sel8", "sel", "node_rec", "in_rec", "out_pad", "acc_lp", "acc_n", "acc_w", "a_origin", "a_dest",
                                           "a_dep", "a_status", "a_order", "cur_lo", "a_dep_sorted", "a_win", "a_ins",
                                           "a_rank")] +
                [("acc_slots", C.c_int64), ("flags", C.c_void_p), ("env_base", C.c_int64), ("due_rate", C.c_float),
                 ("reserved_", C.c_int32), ("bufs_dev", C.c_void_p)])


FLAG_COUNT_AT_NMAX, FLAG_AMBIGUOUS_EDGES, FLAG_PACK_RANGE, FLAG_CHOICE_OVERFLOW = 1, 2, 4, 8


_lib = None
_lock = threading.Lock()


class TarlError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load libtarl_hip.so once; raise (never fall back) when it or one of its symbols is missing."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise TarlError(f"{LIB_PATH} not found: the HIP extension is not built "
                            "(run `python __graft_entry__.py build`); there is no CPU fallback")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as exc:
                raise TarlError(f"{LIB_PATH} does not export {name}; rebuild the extension") from exc
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().tarl_last_error()
        raise TarlError(f"libtarl_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t):
    """Device (or host) address of a tensor, ``None`` -> NULL."""
    return None if t is None else t.data_ptr()


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
