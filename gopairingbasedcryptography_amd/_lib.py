"""ctypes loader for the C-ABI library (include/gpbc_bn254.h).  No fallback: if the HIP library is
missing or no gfx950 device can be bound, every compute call raises."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# GPBC_LIB_PATH: load another build of the same C ABI (kernel-tuning experiments); default is the in-tree library
LIB_PATH = os.environ.get("GPBC_LIB_PATH") or os.path.join(HERE, "libgpbc_bn254.so")

EXPORTS = [
    "gpbc_init", "gpbc_init_devices", "gpbc_num_devices", "gpbc_device_at", "gpbc_set_device", "gpbc_get_device",
    "gpbc_set_host_sharding", "gpbc_shutdown", "gpbc_last_error", "gpbc_device_count", "gpbc_abi_version",
    "gpbc_comm_init_all", "gpbc_comm_get_unique_id", "gpbc_comm_init_rank", "gpbc_comm_ranks", "gpbc_comm_rank", "gpbc_comm_destroy",
    "gpbc_allgather_dev", "gpbc_allgather_all_dev",
    "gpbc_g1_scalar_mul_sum", "gpbc_g2_scalar_mul_sum", "gpbc_g1_scalar_mul_sum_dev", "gpbc_g2_scalar_mul_sum_dev", "gpbc_msm_stats",
    "gpbc_pair_batch", "gpbc_pair_batch_dev", "gpbc_multi_pair", "gpbc_multi_pair_workspace_bytes",
    "gpbc_multi_pair_dev", "gpbc_check_segments_dev", "gpbc_multi_pair_hostseg_dev", "gpbc_set_multi_pair_chunk", "gpbc_multi_pair_fixed_q", "gpbc_multi_pair_fixed_q_dev", "gpbc_pairing_check", "gpbc_miller_loop_dev", "gpbc_final_exp_dev",
    "gpbc_miller_loop", "gpbc_final_exp",
    "gpbc_g1_scalar_mul_batch", "gpbc_g1_scalar_mul_batch_dev", "gpbc_g2_scalar_mul_batch",
    "gpbc_g2_scalar_mul_batch_dev", "gpbc_g1_sum", "gpbc_g2_sum", "gpbc_sum_workspace_bytes",
    "gpbc_g1_sum_dev", "gpbc_g2_sum_dev",
    "gpbc_gt_exp_batch", "gpbc_gt_exp_batch_dev", "gpbc_gt_mul_batch", "gpbc_gt_div_batch",
    "gpbc_gt_inverse_batch", "gpbc_gt_mul_batch_dev", "gpbc_gt_div_batch_dev", "gpbc_gt_inverse_batch_dev",
    "gpbc_fp_mul_batch", "gpbc_profile_begin", "gpbc_profile_end",
    "gpbc_g1_marshal_batch", "gpbc_g2_marshal_batch", "gpbc_gt_marshal_batch",
    "gpbc_g1_marshal_batch_dev", "gpbc_g2_marshal_batch_dev", "gpbc_gt_marshal_batch_dev",
    "gpbc_g1_unmarshal_batch", "gpbc_g2_unmarshal_batch", "gpbc_gt_unmarshal_batch",
    "gpbc_g1_unmarshal_batch_dev", "gpbc_g2_unmarshal_batch_dev", "gpbc_gt_unmarshal_batch_dev",
    "gpbc_g1_map_to_curve_batch", "gpbc_g2_map_to_curve_batch",
    "gpbc_g1_map_to_curve_batch_dev", "gpbc_g2_map_to_curve_batch_dev",
    "gpbc_set_pipelined_miller", "gpbc_set_latency_path", "gpbc_debug_stale_table_once", "gpbc_valu_probe", "gpbc_release_workspaces", "gpbc_hash_to_g1", "gpbc_hash_to_g2", "gpbc_hash_to_field", "gpbc_hash_to_g1_dev", "gpbc_hash_to_g2_dev", "gpbc_hash_to_field_dev",
    "gpbc_fixed_base_table_bytes", "gpbc_g1_fixed_base_create", "gpbc_g2_fixed_base_create", "gpbc_fixed_base_create_dev",
    "gpbc_fixed_base_msm", "gpbc_fixed_base_msm_workspace_bytes", "gpbc_fixed_base_msm_dev", "gpbc_fixed_base_destroy",
]

_lib = None


class EngineError(RuntimeError):
    """Raised when the C ABI returns a negative gpbc_status."""


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(
                "HIP extension %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and device tensors /
        # streams come from it, so load torch first and let the dynamic loader resolve this library's
        # libamdhip64.so.7 dependency to that already-loaded copy (two runtimes cannot share the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        lib.gpbc_last_error.restype = ctypes.c_char_p
        lib.gpbc_multi_pair_workspace_bytes.restype = ctypes.c_size_t
        lib.gpbc_multi_pair_workspace_bytes.argtypes = [ctypes.c_size_t, ctypes.c_size_t]
        lib.gpbc_sum_workspace_bytes.restype = ctypes.c_size_t
        lib.gpbc_sum_workspace_bytes.argtypes = [ctypes.c_size_t, ctypes.c_int]
        lib.gpbc_fixed_base_table_bytes.restype = ctypes.c_size_t
        lib.gpbc_fixed_base_table_bytes.argtypes = [ctypes.c_size_t, ctypes.c_int]
        lib.gpbc_fixed_base_msm_workspace_bytes.restype = ctypes.c_size_t
        lib.gpbc_fixed_base_msm_workspace_bytes.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
        lib.gpbc_set_latency_path.argtypes = [ctypes.c_long]
        _lib = lib
    return _lib


def check(rc):
    if rc < 0:
        raise EngineError("gpbc error %d: %s" % (rc, load().gpbc_last_error().decode()))
    return rc
