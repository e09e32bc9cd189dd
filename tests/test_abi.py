"""CPU: the C-ABI library loads and exports every symbol include/tarl_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "tarl_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tarl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from tarl_hip import lib
    assert os.path.exists(lib.LIB_PATH), "build the extension first: python __graft_entry__.py build"
    so = ctypes.CDLL(lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(so, n), f"{n} declared in include/tarl_hip.h but not exported"
    # the ctypes table binds exactly the declared entry points
    assert sorted(lib.SIGNATURES) == names
    L = lib.load()
    assert L.tarl_abi_version() == 5
    hdr = open(os.path.join(ROOT, "include", "tarl_hip.h")).read()
    assert int(re.search(r"#define TARL_ABI_VERSION (\d+)", hdr).group(1)) == 5      # header, library and this test move together


def test_plan_create_rejects_bad_input_without_gpu_compute():
    """Argument validation happens on the host before any HIP call."""
    import torch
    from tarl_hip import lib
    L = lib.load()
    ei = torch.tensor([[0, 1], [1, 5]], dtype=torch.int64)   # node 5 out of range for 3 nodes
    h = ctypes.c_void_p()
    rc = L.tarl_plan_create(ei.data_ptr(), 2, 3, None, ctypes.byref(h))
    assert rc == -1 and b"out of range" in L.tarl_last_error()


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from tarl_hip import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(lib.TarlError):
        lib.load()


def test_cpu_tensor_is_refused():
    """No CPU fallback: a host tensor handed to an op raises instead of being computed elsewhere."""
    import pytest
    import torch
    from tarl_hip import lib, ops
    with pytest.raises(lib.TarlError):
        ops._check_dev(torch.zeros(4), torch.float32, "x")


def test_entry_points_reject_bad_arguments_on_the_host():
    """Every entry point validates its arguments before the first HIP call and reports through the error code +
    tarl_last_error (no exceptions cross the ABI): exercised here without a GPU."""
    import ctypes as C
    from tarl_hip import lib
    L = lib.load()
    null = None
    assert L.tarl_rollout_env_supported(null) == 0
    assert L.tarl_apsp_scratch_bytes(null, 1) == -1
    assert L.tarl_apsp(null, null, 1, 0, null, 0, null, null, null) == -1
    assert b"null" in L.tarl_last_error()
    assert L.tarl_fused_rollout(null, null, 1, 15, 1, null, 0.0, null, null, null, 0, 0, null, 1, 9, null, null, 0.0, 0, 0,
                                0, null, null, null, null, null, null, null, null, null, 0, null, null, null, null) == -1
    assert L.tarl_fused_rollout_scratch_ints(null, 1, 1) == -1
    assert L.tarl_rollout_env(null, null, 1, 15, 1, null, 0.0, null, null, null, 0, 0, null, 1, 9, null, null, 0.0, 0, 0, 0,
                              null, null, null, null, null, null, null, 0, null, null, null, null) == -1
    assert L.tarl_rollout_gather(null, null, null, 1, 1, 1, null, 0, null, null, null) == -1
    assert L.tarl_fused_apply_choice(null, null, 1, null, null) == -1
    assert L.tarl_value_mpnn_fwd(null, null, 1, null, null, 0, null, null, null, null, null, null) == -1
    assert L.tarl_select_next_hop(null, 1, 0, 52, 15, 4, null, 1, 9, null, 0, null) == -1
    # round 5's entry points: the noise export, the scratch sizes of the chunked backward kernels, the pointer table
    assert L.tarl_noise_export(null, 0, 1, 1, null, 1, null, null) == -1 and b"null" in L.tarl_last_error()
    assert L.tarl_critic_mlp_bwd_scratch_floats(0, 10) == -1 and L.tarl_critic_mlp_bwd_scratch_floats(32, 10) == 2 * 32 * 64
    assert L.tarl_critic_mlp_bwd_scratch_floats(1024, 10) > 2 * 1024 * 64          # partial sums of the row chunks
    assert L.tarl_policy_edge_logits_bwd_scratch_floats(null, 8) == -1
    assert L.tarl_fused_bufs_bytes() >= 2 * 27 * 8
    ms_all, ms_late, n = (C.c_double * 3)(), (C.c_double * 3)(), (C.c_int64 * 2)()
    assert L.tarl_prof_collect(0, ms_all, ms_late, n) == 0 and n[0] == 0 and n[1] == 0


def test_driver_build_entry_accepts_the_built_library():
    """__graft_entry__.build() is the driver's "does it build" check: after a build it must accept the library this tree
    produces (round 5 shipped an entry point that still expected the previous ABI number until its last hour)."""
    import __graft_entry__ as g
    src = open(g.__file__).read()
    assert "TARL_ABI_VERSION" in src and "tarl_abi_version() ==" in src      # compares against the header, not a literal
    from tarl_hip import lib
    hdr = open(os.path.join(ROOT, "include", "tarl_hip.h")).read()
    assert lib.load().tarl_abi_version() == int(re.search(r"#define TARL_ABI_VERSION (\d+)", hdr).group(1))
