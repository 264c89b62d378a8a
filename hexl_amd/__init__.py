"""hexl_amd -- MI355X-native NTT / Eltwise hot path of intel/hexl.

Python host-side mirror of the reference's operator interface
(``intel::hexl::NTT``, ``intel::hexl::Eltwise*``; hexl/include/hexl/ntt/ntt.hpp,
hexl/include/hexl/eltwise/*.hpp) over the C-ABI of ``include/hexl_amd.h``.
Names, argument meaning and error behaviour follow the reference; operands are
device-resident ``torch`` tensors (``torch.int64`` / ``torch.uint64`` storage
holding uint64 words), and a leading batch dimension is allowed everywhere a
polynomial is expected.  PyTorch is used for device memory and streams only.

There is no CPU fallback: importing works without a GPU (so the library's
symbols can be checked), but every compute call raises ``HexlAmdError`` unless
the HIP kernels can run.
"""
import ctypes as C
import os

try:
    # torch bundles its own libamdhip64.so.7; it has to be the first HIP runtime
    # mapped into the process, or two runtimes end up half-initialised.
    import torch as _torch_first  # noqa: F401
except ImportError:  # pure ctypes use without torch: the system ROCm runtime
    _torch_first = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEXL_AMD_LIB: developer override to A/B an experimental build of the same C-ABI
_LIB_PATH = os.environ.get("HEXL_AMD_LIB") or os.path.join(_HERE, "lib", "libhexl_amd.so")

__all__ = [
    "NTT", "EltwiseAddMod", "EltwiseSubMod", "EltwiseMultMod", "EltwiseFMAMod",
    "EltwiseReduceMod", "EltwiseReduceFMAMod", "EltwiseCmpAdd", "EltwiseCmpSubMod", "CMPINT", "DyadicMultiply", "KeySwitch",
    "KeySwitchBatch", "DyadicMultiplyBatch",
    "HexlAmdError", "lib", "LIB_PATH",
    "MinimalPrimitiveRoot", "GeneratePrimes", "IsPrime", "InverseMod", "MultiplyMod",
    "PowMod", "IsPrimitiveRoot", "ReverseBits", "MultiplyFactor", "fill_splitmix",
    "from_numpy", "to_numpy",
]

LIB_PATH = _LIB_PATH


class HexlAmdError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIB_PATH):
        raise HexlAmdError(
            f"{_LIB_PATH} is missing: build it with `python -m hexl_amd.build` "
            "(hipcc, gfx950). hexl_amd has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    u64, p64, vp, ci = C.c_uint64, C.c_void_p, C.c_void_p, C.c_int

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("hexl_amd_last_error", C.c_char_p)
    sig("hexl_amd_device_count", ci, C.POINTER(ci))
    sig("hexl_amd_pointer_is_device", ci, vp)
    sig("hexl_amd_ntt_create", ci, C.POINTER(vp), u64, u64, u64, ci)
    sig("hexl_amd_ntt_destroy", ci, vp)
    sig("hexl_amd_ntt_degree", u64, vp)
    sig("hexl_amd_ntt_modulus", u64, vp)
    sig("hexl_amd_ntt_root_of_unity", u64, vp)
    sig("hexl_amd_ntt_device", ci, vp)
    sig("hexl_amd_ntt_table", C.POINTER(u64), vp, ci)
    for name in ("hexl_amd_ntt_forward", "hexl_amd_ntt_inverse"):
        sig(name, ci, vp, p64, p64, u64, u64, u64, vp)
    for name in ("hexl_amd_ntt_forward_rns", "hexl_amd_ntt_inverse_rns"):
        sig(name, ci, C.POINTER(vp), u64, p64, p64, u64, u64, u64, vp)
    for name in ("hexl_amd_ntt_forward_host", "hexl_amd_ntt_inverse_host"):
        sig(name, ci, vp, p64, p64, u64, u64, u64)
    sig("hexl_amd_eltwise_add_mod", ci, p64, p64, p64, u64, u64, vp)
    sig("hexl_amd_eltwise_add_mod_scalar", ci, p64, p64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_sub_mod", ci, p64, p64, p64, u64, u64, vp)
    sig("hexl_amd_eltwise_sub_mod_scalar", ci, p64, p64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_mult_mod", ci, p64, p64, p64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_fma_mod", ci, p64, p64, u64, p64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_reduce_mod", ci, p64, p64, u64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_reduce_fma_mod", ci, p64, p64, u64, p64, u64, u64, u64, vp)
    sig("hexl_amd_eltwise_host", ci, ci, p64, p64, p64, u64, u64, u64, u64, u64)
    sig("hexl_amd_dyadic_multiply", ci, p64, p64, p64, u64, C.POINTER(u64), u64, vp)
    sig("hexl_amd_dyadic_multiply_batch", ci, p64, p64, p64, u64, u64, C.POINTER(u64), u64, vp)
    sig("hexl_amd_dyadic_multiply_host", ci, p64, p64, p64, u64, C.POINTER(u64), u64)
    sig("hexl_amd_key_switch", ci, p64, p64, u64, u64, u64, u64, u64, C.POINTER(u64),
        C.POINTER(vp), C.POINTER(u64), vp)
    sig("hexl_amd_key_switch_batch", ci, p64, p64, u64, u64, u64, u64, u64, u64, C.POINTER(u64),
        C.POINTER(vp), C.POINTER(u64), vp)
    sig("hexl_amd_key_switch_host", ci, p64, p64, u64, u64, u64, u64, u64, C.POINTER(u64),
        C.POINTER(vp), C.POINTER(u64))
    sig("hexl_amd_eltwise_cmp_add", ci, p64, p64, u64, ci, u64, u64, vp)
    sig("hexl_amd_eltwise_cmp_sub_mod", ci, p64, p64, u64, u64, ci, u64, u64, vp)
    sig("hexl_amd_eltwise_cmp_host", ci, p64, p64, u64, u64, ci, u64, u64)
    sig("hexl_amd_multiply_factor", u64, u64, u64, u64)
    sig("hexl_amd_inverse_mod", u64, u64, u64)
    sig("hexl_amd_multiply_mod", u64, u64, u64, u64)
    sig("hexl_amd_pow_mod", u64, u64, u64, u64)
    sig("hexl_amd_is_primitive_root", ci, u64, u64, u64)
    sig("hexl_amd_generate_primitive_root", u64, u64, u64)
    sig("hexl_amd_minimal_primitive_root", u64, u64, u64)
    sig("hexl_amd_reverse_bits", u64, u64, u64)
    sig("hexl_amd_is_prime", ci, u64)
    sig("hexl_amd_generate_primes", C.c_size_t, C.POINTER(u64), C.c_size_t, C.c_size_t, ci,
        C.c_size_t)
    sig("hexl_amd_ntt_check_arguments", ci, u64, u64)
    sig("hexl_amd_fill_splitmix", ci, p64, u64, u64, u64, u64, vp)
    sig("hexl_amd_profile_start", ci, ci)
    sig("hexl_amd_profile_stop", ci, C.POINTER(ci))
    sig("hexl_amd_profile_get", ci, ci, C.POINTER(C.c_char_p), C.POINTER(C.c_float))
    sig("hexl_amd_set_tuning", ci, C.c_char_p, u64)
    sig("hexl_amd_get_counter", ci, C.c_char_p, C.POINTER(u64))
    sig("hexl_amd_ntt_forward_map", ci, C.POINTER(vp), u64, C.POINTER(C.c_uint8), u64, u64, p64, p64,
        u64, u64, u64, vp)
    sig("hexl_amd_ntt_inverse_map", ci, C.POINTER(vp), u64, C.POINTER(C.c_uint8), u64, u64, p64, p64,
        u64, u64, u64, vp)
    sig("hexl_amd_ntt_forward_indexed", ci, C.POINTER(vp), u64, C.POINTER(C.c_uint32), p64, p64, u64,
        u64, u64, vp)
    sig("hexl_amd_ntt_inverse_indexed", ci, C.POINTER(vp), u64, C.POINTER(C.c_uint32), p64, p64, u64,
        u64, u64, vp)
    sig("hexl_amd_host_alloc", ci, C.POINTER(vp), u64)
    sig("hexl_amd_host_free", ci, vp)
    sig("hexl_amd_host_register", ci, vp, u64)
    sig("hexl_amd_host_unregister", ci, vp)
    sig("hexl_amd_pointer_kind", ci, vp)
    sig("hexl_amd_check_bounds", ci, p64, u64, u64, C.POINTER(u64))
    sig("hexl_amd_device_alloc", ci, C.POINTER(vp), u64, ci)
    sig("hexl_amd_device_free", ci, vp)
    sig("hexl_amd_copy", ci, vp, vp, u64, vp, ci)
    sig("hexl_amd_synchronize", ci, vp)
    sig("hexl_amd_set_device", ci, ci)
    sig("hexl_amd_get_device", ci, C.POINTER(ci))
    sig("hexl_amd_stream_create", ci, C.POINTER(vp), ci)
    sig("hexl_amd_stream_destroy", ci, vp)
    sig("hexl_amd_release_stream_workspaces", ci, vp)
    sig("hexl_amd_release_workspaces", ci)
    return lib


lib = _load()

# Every symbol include/hexl_amd.h declares (checked by tests/test_capi_symbols.py)
C_ABI_SYMBOLS = [
    "hexl_amd_last_error", "hexl_amd_device_count", "hexl_amd_pointer_is_device",
    "hexl_amd_ntt_create",
    "hexl_amd_ntt_destroy", "hexl_amd_ntt_degree", "hexl_amd_ntt_modulus",
    "hexl_amd_ntt_root_of_unity", "hexl_amd_ntt_device", "hexl_amd_ntt_table",
    "hexl_amd_ntt_forward", "hexl_amd_ntt_inverse", "hexl_amd_ntt_forward_rns",
    "hexl_amd_ntt_inverse_rns", "hexl_amd_ntt_forward_host", "hexl_amd_ntt_inverse_host",
    "hexl_amd_eltwise_add_mod", "hexl_amd_eltwise_add_mod_scalar", "hexl_amd_eltwise_sub_mod",
    "hexl_amd_eltwise_sub_mod_scalar", "hexl_amd_eltwise_mult_mod", "hexl_amd_eltwise_fma_mod",
    "hexl_amd_eltwise_reduce_mod", "hexl_amd_eltwise_reduce_fma_mod", "hexl_amd_eltwise_host",
    "hexl_amd_eltwise_cmp_add", "hexl_amd_eltwise_cmp_sub_mod", "hexl_amd_eltwise_cmp_host",
    "hexl_amd_dyadic_multiply", "hexl_amd_dyadic_multiply_batch", "hexl_amd_dyadic_multiply_host",
    "hexl_amd_key_switch", "hexl_amd_key_switch_batch", "hexl_amd_key_switch_host",
    "hexl_amd_multiply_factor", "hexl_amd_inverse_mod", "hexl_amd_multiply_mod",
    "hexl_amd_pow_mod", "hexl_amd_is_primitive_root", "hexl_amd_generate_primitive_root",
    "hexl_amd_minimal_primitive_root", "hexl_amd_reverse_bits", "hexl_amd_is_prime",
    "hexl_amd_generate_primes", "hexl_amd_ntt_check_arguments", "hexl_amd_fill_splitmix",
    "hexl_amd_profile_start", "hexl_amd_profile_stop", "hexl_amd_profile_get",
    "hexl_amd_set_tuning", "hexl_amd_get_counter",
    "hexl_amd_ntt_forward_map", "hexl_amd_ntt_inverse_map", "hexl_amd_ntt_forward_indexed",
    "hexl_amd_ntt_inverse_indexed", "hexl_amd_release_stream_workspaces",
    "hexl_amd_release_workspaces", "hexl_amd_host_alloc", "hexl_amd_host_free",
    "hexl_amd_host_register", "hexl_amd_host_unregister", "hexl_amd_pointer_kind",
    "hexl_amd_check_bounds", "hexl_amd_device_alloc", "hexl_amd_device_free", "hexl_amd_copy",
    "hexl_amd_synchronize", "hexl_amd_set_device", "hexl_amd_get_device",
    "hexl_amd_stream_create", "hexl_amd_stream_destroy",
]


def _check(rc):
    if rc != 0:
        raise HexlAmdError(lib.hexl_amd_last_error().decode() or f"hexl_amd error {rc}")


def _torch():
    import torch
    return torch


def _require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise HexlAmdError("no MI355X visible: hexl_amd has no CPU fallback")
    return torch


def _ptr(t, need=0, device=None):
    """Device pointer of tensor `t`, which must hold at least `need` words and live on
    `device` (default: the current device, where the kernels are launched)."""
    torch = _torch()
    if not isinstance(t, torch.Tensor):
        raise HexlAmdError("expected a torch tensor")
    if not t.is_cuda:
        raise HexlAmdError("expected a device (cuda) tensor; host tensors go through the "
                           "C-ABI *_host entry points")
    if t.dtype not in (torch.int64, torch.uint64):
        raise HexlAmdError(f"expected int64/uint64 storage, got {t.dtype}")
    if not t.is_contiguous():
        raise HexlAmdError("expected a contiguous tensor")
    if t.numel() < need:
        raise HexlAmdError(f"tensor holds {t.numel()} words, the call needs {need}")
    want = torch.cuda.current_device() if device is None else device
    if t.device.index != want:
        raise HexlAmdError(f"tensor is on cuda:{t.device.index}, the call runs on cuda:{want}")
    return C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def from_numpy(a, device="cuda"):
    """uint64 numpy array -> device tensor (int64 storage, same bits).  The bytes travel through
    hexl_amd_copy -- the library's pinned slots -- not through torch's copy of pageable memory,
    which from about 1 MiB lets the HIP runtime pin the array's pages on the fly (the path the
    GPU memory access faults of rounds 4 and 5 sat in: EXPERIMENTS.md section 10)."""
    import numpy as np
    torch = _require_gpu()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    t = torch.empty(a.shape, dtype=torch.int64, device=device)
    if a.size:
        with torch.cuda.device(t.device):
            _check(lib.hexl_amd_copy(C.c_void_p(t.data_ptr()), a.ctypes.data_as(C.c_void_p), a.nbytes,
                                     _stream(), 1))
    return t


def to_numpy(t):
    """device tensor -> uint64 numpy array (through hexl_amd_copy, see from_numpy)."""
    import numpy as np
    torch = _torch()
    if t.dtype == torch.uint64:
        t = t.view(torch.int64)
    if not t.is_cuda:
        return t.detach().numpy().view(np.uint64)
    t = t.detach().contiguous()
    out = np.empty(tuple(t.shape), dtype=np.uint64)
    if out.size:
        with torch.cuda.device(t.device):
            _check(lib.hexl_amd_copy(out.ctypes.data_as(C.c_void_p), C.c_void_p(t.data_ptr()), out.nbytes,
                                     _stream(), 1))
    return out


# ----------------------------------------------------------------------------
# number theory (host) -- hexl/include/hexl/number-theory/number-theory.hpp
# ----------------------------------------------------------------------------
def MultiplyFactor(operand, bit_shift, modulus):
    return lib.hexl_amd_multiply_factor(operand, bit_shift, modulus)


def InverseMod(x, modulus):
    return lib.hexl_amd_inverse_mod(x, modulus)


def MultiplyMod(x, y, modulus):
    return lib.hexl_amd_multiply_mod(x, y, modulus)


def PowMod(base, exp, modulus):
    return lib.hexl_amd_pow_mod(base, exp, modulus)


def IsPrimitiveRoot(root, degree, modulus):
    return bool(lib.hexl_amd_is_primitive_root(root, degree, modulus))


def MinimalPrimitiveRoot(degree, modulus):
    return lib.hexl_amd_minimal_primitive_root(degree, modulus)


def ReverseBits(x, bit_width):
    return lib.hexl_amd_reverse_bits(x, bit_width)


def IsPrime(n):
    return bool(lib.hexl_amd_is_prime(n))


def GeneratePrimes(num_primes, bit_size, prefer_small_primes, ntt_size=1):
    out = (C.c_uint64 * num_primes)()
    found = lib.hexl_amd_generate_primes(out, num_primes, bit_size,
                                         int(bool(prefer_small_primes)), ntt_size)
    if found != num_primes:
        raise HexlAmdError("Failed to find enough primes")
    return [int(x) for x in out]


# ----------------------------------------------------------------------------
# NTT -- hexl/include/hexl/ntt/ntt.hpp:22-293
# ----------------------------------------------------------------------------
class NTT:
    """Negacyclic forward / inverse NTT plan resident on one GPU.

    Mirrors ``intel::hexl::NTT``: ``NTT(degree, q)`` or
    ``NTT(degree, q, root_of_unity)``.  ``ComputeForward(result, operand,
    input_mod_factor, output_mod_factor)`` and ``ComputeInverse`` take device
    tensors of ``k * degree`` words (k >= 1 polynomials, back to back);
    ``result`` may be ``operand``.
    """

    def __init__(self, degree, q, root_of_unity=0, device=None):
        torch = _require_gpu()
        if device is None:
            device = torch.cuda.current_device()
        h = C.c_void_p()
        _check(lib.hexl_amd_ntt_create(C.byref(h), degree, q, root_of_unity, int(device)))
        self._h = h
        self._device = int(device)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and lib is not None:  # `lib` is already gone at interpreter shutdown
            lib.hexl_amd_ntt_destroy(h)

    @staticmethod
    def CheckArguments(degree, modulus):
        return bool(lib.hexl_amd_ntt_check_arguments(degree, modulus))

    def GetDevice(self):
        return self._device

    def GetDegree(self):
        return lib.hexl_amd_ntt_degree(self._h)

    def GetModulus(self):
        return lib.hexl_amd_ntt_modulus(self._h)

    def GetMinimalRootOfUnity(self):
        return lib.hexl_amd_ntt_root_of_unity(self._h)

    def _table(self, which):
        import numpy as np
        n = self.GetDegree()
        p = lib.hexl_amd_ntt_table(self._h, which)
        return np.ctypeslib.as_array(p, shape=(n,)).copy()

    def GetRootOfUnityPowers(self):
        return self._table(0)

    def GetPrecon32RootOfUnityPowers(self):
        return self._table(1)

    def GetPrecon64RootOfUnityPowers(self):
        return self._table(2)

    def GetInvRootOfUnityPowers(self):
        return self._table(3)

    def GetPrecon32InvRootOfUnityPowers(self):
        return self._table(4)

    def GetPrecon52InvRootOfUnityPowers(self):
        return self._table(5)

    def GetPrecon64InvRootOfUnityPowers(self):
        return self._table(6)

    def _batch(self, result, operand):
        n = self.GetDegree()
        if operand.numel() % n or result.numel() != operand.numel():
            raise HexlAmdError("operand/result must hold k * degree words")
        return operand.numel() // n

    def ComputeForward(self, result, operand, input_mod_factor, output_mod_factor):
        b = self._batch(result, operand)
        dev = self.GetDevice()
        _check(lib.hexl_amd_ntt_forward(self._h, _ptr(result, 0, dev), _ptr(operand, 0, dev), b,
                                        input_mod_factor, output_mod_factor, _stream()))

    def ComputeInverse(self, result, operand, input_mod_factor, output_mod_factor):
        b = self._batch(result, operand)
        dev = self.GetDevice()
        _check(lib.hexl_amd_ntt_inverse(self._h, _ptr(result, 0, dev), _ptr(operand, 0, dev), b,
                                        input_mod_factor, output_mod_factor, _stream()))


def _rns(fn, plans, result, operand, in_mf, out_mf):
    k = len(plans)
    n = plans[0].GetDegree()
    if operand.numel() % (k * n) or result.numel() != operand.numel():
        raise HexlAmdError("operand must hold len(plans) * batch * degree words")
    arr = (C.c_void_p * k)(*[p._h for p in plans])
    dev = plans[0].GetDevice()
    _check(fn(arr, k, _ptr(result, 0, dev), _ptr(operand, 0, dev), operand.numel() // (k * n), in_mf, out_mf,
              _stream()))


def ComputeForwardRNS(plans, result, operand, input_mod_factor, output_mod_factor):
    """plans[k] transforms polynomials [k*B, (k+1)*B) of the (K, B, N) operand."""
    _rns(lib.hexl_amd_ntt_forward_rns, plans, result, operand, input_mod_factor,
         output_mod_factor)


def ComputeInverseRNS(plans, result, operand, input_mod_factor, output_mod_factor):
    _rns(lib.hexl_amd_ntt_inverse_rns, plans, result, operand, input_mod_factor,
         output_mod_factor)


def _map(fn, plans, plan_of_slot, inner, result, operand, in_mf, out_mf):
    k = len(plans)
    n = plans[0].GetDegree()
    if operand.numel() % n or result.numel() != operand.numel():
        raise HexlAmdError("operand must hold whole polynomials")
    arr = (C.c_void_p * k)(*[p._h for p in plans])
    tab = (C.c_uint8 * len(plan_of_slot))(*plan_of_slot)
    dev = plans[0].GetDevice()
    _check(fn(arr, k, tab, len(plan_of_slot), inner, _ptr(result, 0, dev), _ptr(operand, 0, dev),
              operand.numel() // n, in_mf, out_mf, _stream()))


def ComputeForwardMap(plans, plan_of_slot, inner, result, operand, input_mod_factor,
                      output_mod_factor):
    """Polynomial i uses plans[plan_of_slot[(i // inner) % len(plan_of_slot)]]: inner = 1 and
    plan_of_slot = range(k) is SEAL's [ciphertext][component][modulus][N] layout."""
    _map(lib.hexl_amd_ntt_forward_map, plans, plan_of_slot, inner, result, operand,
         input_mod_factor, output_mod_factor)


def ComputeInverseMap(plans, plan_of_slot, inner, result, operand, input_mod_factor,
                      output_mod_factor):
    _map(lib.hexl_amd_ntt_inverse_map, plans, plan_of_slot, inner, result, operand,
         input_mod_factor, output_mod_factor)


def _indexed(fn, plans, prime_index, result, operand, in_mf, out_mf):
    k = len(plans)
    n = plans[0].GetDegree()
    polys = operand.numel() // n
    if operand.numel() % n or result.numel() != operand.numel() or len(prime_index) != polys:
        raise HexlAmdError("one prime index per polynomial")
    arr = (C.c_void_p * k)(*[p._h for p in plans])
    idx = (C.c_uint32 * polys)(*[int(v) for v in prime_index])
    dev = plans[0].GetDevice()
    _check(fn(arr, k, idx, _ptr(result, 0, dev), _ptr(operand, 0, dev), polys, in_mf, out_mf,
              _stream()))


def ComputeForwardIndexed(plans, prime_index, result, operand, input_mod_factor,
                          output_mod_factor):
    """Polynomial i uses plans[prime_index[i]]."""
    _indexed(lib.hexl_amd_ntt_forward_indexed, plans, prime_index, result, operand,
             input_mod_factor, output_mod_factor)


def ComputeInverseIndexed(plans, prime_index, result, operand, input_mod_factor,
                          output_mod_factor):
    _indexed(lib.hexl_amd_ntt_inverse_indexed, plans, prime_index, result, operand,
             input_mod_factor, output_mod_factor)


# ----------------------------------------------------------------------------
# Eltwise -- hexl/include/hexl/eltwise/*.hpp
# ----------------------------------------------------------------------------
def EltwiseAddMod(result, operand1, operand2, n, modulus):
    """operand2: tensor (vector-vector) or int (vector-scalar)."""
    if isinstance(operand2, int):
        _check(lib.hexl_amd_eltwise_add_mod_scalar(_ptr(result, n), _ptr(operand1, n), operand2, n,
                                                   modulus, _stream()))
    else:
        _check(lib.hexl_amd_eltwise_add_mod(_ptr(result, n), _ptr(operand1, n), _ptr(operand2, n), n,
                                            modulus, _stream()))


def EltwiseSubMod(result, operand1, operand2, n, modulus):
    if isinstance(operand2, int):
        _check(lib.hexl_amd_eltwise_sub_mod_scalar(_ptr(result, n), _ptr(operand1, n), operand2, n,
                                                   modulus, _stream()))
    else:
        _check(lib.hexl_amd_eltwise_sub_mod(_ptr(result, n), _ptr(operand1, n), _ptr(operand2, n), n,
                                            modulus, _stream()))


def EltwiseMultMod(result, operand1, operand2, n, modulus, input_mod_factor):
    _check(lib.hexl_amd_eltwise_mult_mod(_ptr(result, n), _ptr(operand1, n), _ptr(operand2, n), n,
                                         modulus, input_mod_factor, _stream()))


def EltwiseFMAMod(result, arg1, arg2, arg3, n, modulus, input_mod_factor):
    p3 = _ptr(arg3, n) if arg3 is not None else None
    _check(lib.hexl_amd_eltwise_fma_mod(_ptr(result, n), _ptr(arg1, n), arg2, p3, n, modulus,
                                        input_mod_factor, _stream()))


def EltwiseReduceMod(result, operand, n, modulus, input_mod_factor, output_mod_factor):
    _check(lib.hexl_amd_eltwise_reduce_mod(_ptr(result, n), _ptr(operand, n), n, modulus,
                                           input_mod_factor, output_mod_factor, _stream()))


def DyadicMultiply(result, operand1, operand2, n, moduli):
    """hexl/include/hexl/experimental/seal/dyadic-multiply.hpp:26-28; `moduli` is a host
    sequence of integers."""
    arr = (C.c_uint64 * len(moduli))(*[int(m) for m in moduli])
    k = n * len(moduli)
    _check(lib.hexl_amd_dyadic_multiply(_ptr(result, 3 * k), _ptr(operand1, 2 * k), _ptr(operand2, 2 * k), n, arr,
                                        len(moduli), _stream()))


def DyadicMultiplyBatch(result, operand1, operand2, num_pairs, n, moduli):
    """num_pairs ciphertext pairs with the same moduli in one launch."""
    arr = (C.c_uint64 * len(moduli))(*[int(m) for m in moduli])
    k = n * len(moduli) * num_pairs
    _check(lib.hexl_amd_dyadic_multiply_batch(_ptr(result, 3 * k), _ptr(operand1, 2 * k),
                                              _ptr(operand2, 2 * k), num_pairs, n, arr,
                                              len(moduli), _stream()))


def KeySwitch(result, t_target_iter, n, decomp_modulus_size, key_modulus_size, rns_modulus_size,
              key_component_count, moduli, k_switch_keys, modswitch_factors):
    """hexl/include/hexl/experimental/seal/key-switch.hpp:40-46.  result, t_target_iter and the
    entries of k_switch_keys are device tensors; moduli / modswitch_factors host sequences."""
    mod = (C.c_uint64 * len(moduli))(*[int(m) for m in moduli])
    msf = (C.c_uint64 * len(modswitch_factors))(*[int(m) for m in modswitch_factors])
    if (len(moduli) < key_modulus_size or len(modswitch_factors) < decomp_modulus_size or
            len(k_switch_keys) < decomp_modulus_size):
        raise HexlAmdError("moduli / modswitch_factors / k_switch_keys are shorter than the sizes say")
    key_words = key_component_count * key_modulus_size * n
    keys = (C.c_void_p * len(k_switch_keys))(*[_ptr(k, key_words).value for k in k_switch_keys])
    _check(lib.hexl_amd_key_switch(_ptr(result, key_component_count * decomp_modulus_size * n),
                                   _ptr(t_target_iter, decomp_modulus_size * n), n, decomp_modulus_size,
                                   key_modulus_size, rns_modulus_size, key_component_count, mod,
                                   keys, msf, _stream()))


def KeySwitchBatch(result, t_target_iter, num_targets, n, decomp_modulus_size, key_modulus_size,
                   rns_modulus_size, key_component_count, moduli, k_switch_keys, modswitch_factors):
    """num_targets ciphertexts with the same keys and moduli in one call (at most nine launches
    whatever the sizes): targets and results back to back in the layouts of KeySwitch."""
    mod = (C.c_uint64 * len(moduli))(*[int(m) for m in moduli])
    msf = (C.c_uint64 * len(modswitch_factors))(*[int(m) for m in modswitch_factors])
    if (len(moduli) < key_modulus_size or len(modswitch_factors) < decomp_modulus_size or
            len(k_switch_keys) < decomp_modulus_size):
        raise HexlAmdError("moduli / modswitch_factors / k_switch_keys are shorter than the sizes say")
    key_words = key_component_count * key_modulus_size * n
    keys = (C.c_void_p * len(k_switch_keys))(*[_ptr(k, key_words).value for k in k_switch_keys])
    _check(lib.hexl_amd_key_switch_batch(
        _ptr(result, num_targets * key_component_count * decomp_modulus_size * n),
        _ptr(t_target_iter, num_targets * decomp_modulus_size * n), num_targets, n,
        decomp_modulus_size, key_modulus_size, rns_modulus_size, key_component_count, mod, keys,
        msf, _stream()))


class CMPINT:
    """hexl/include/hexl/util/util.hpp:16-25."""
    EQ, LT, LE, FALSE, NE, NLT, NLE, TRUE = range(8)


def EltwiseCmpAdd(result, operand1, n, cmp, bound, diff):
    """hexl/include/hexl/eltwise/eltwise-cmp-add.hpp:24-25."""
    _check(lib.hexl_amd_eltwise_cmp_add(_ptr(result, n), _ptr(operand1, n), n, int(cmp), bound, diff,
                                        _stream()))


def EltwiseCmpSubMod(result, operand1, n, modulus, cmp, bound, diff):
    """hexl/include/hexl/eltwise/eltwise-cmp-sub-mod.hpp:26-28."""
    _check(lib.hexl_amd_eltwise_cmp_sub_mod(_ptr(result, n), _ptr(operand1, n), n, modulus, int(cmp),
                                            bound, diff, _stream()))


def EltwiseReduceFMAMod(result, arg1, arg2, arg3, n, modulus, input_mod_factor):
    """Fused EltwiseReduceMod(q -> 1) + EltwiseFMAMod (BASELINE config 5)."""
    p3 = _ptr(arg3, n) if arg3 is not None else None
    _check(lib.hexl_amd_eltwise_reduce_fma_mod(_ptr(result, n), _ptr(arg1, n), arg2, p3, n, modulus,
                                               input_mod_factor, _stream()))


def profile_start(max_records=4096):
    _check(lib.hexl_amd_profile_start(max_records))


def profile_stop():
    """Returns [(kernel_name, milliseconds), ...] in launch order."""
    n = C.c_int(0)
    _check(lib.hexl_amd_profile_stop(C.byref(n)))
    out = []
    for i in range(n.value):
        name, ms = C.c_char_p(), C.c_float()
        _check(lib.hexl_amd_profile_get(i, C.byref(name), C.byref(ms)))
        out.append((name.value.decode(), ms.value))
    return out


def get_counter(key):
    """Process-wide event counters: "ks_graph_captures", "ks_graph_replays", "ks_eager", "host_polls",
    "host_poll_timeouts"."""
    v = C.c_uint64(0)
    _check(lib.hexl_amd_get_counter(key.encode(), C.byref(v)))
    return int(v.value)


def set_tuning(key, value):
    """Tuning knobs (include/hexl_amd.h documents them): "fp64", "fp64_long", "lazy_family", "h60" (read when a
    plan is created), "tile13", "bigtile", "walk14", "host_bounce_kb", "host_direct_copy", "host_copy_threads", "host_poll", "ks_graph", "ks_fuse", "ks_mac_onestep".
    The library reads no environment variable; results never depend on the knobs."""
    _check(lib.hexl_amd_set_tuning(key.encode(), int(value)))


def fill_splitmix(data, n, batch, seed0, bound):
    """Device-side synthetic input: poly b = splitmix64(seed0 + b) mod bound."""
    _check(lib.hexl_amd_fill_splitmix(_ptr(data, n * batch), n, batch, seed0, bound, _stream()))
