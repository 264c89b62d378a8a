"""ctypes binding of the CPU oracle (oracle/hexl_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, bench.py's ``cpu_baseline``
leg and ``__graft_entry__.smoke()``; never from the ``hexl_amd`` package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libhexl_oracle.so")

u64 = C.c_uint64
p64 = C.POINTER(C.c_uint64)


def build(force=False):
    """Compile libhexl_oracle.so with the committed Makefile (gcc only)."""
    src = os.path.join(_HERE, "hexl_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def _load():
    build()
    lib = C.CDLL(_LIB_PATH)

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("ho_mul_hi64", u64, u64, u64)
    sig("ho_reduce128", u64, u64, u64, u64)
    sig("ho_divide_u128_u64_lo", u64, u64, u64, u64)
    sig("ho_msb", u64, u64)
    sig("ho_multiply_factor", u64, u64, u64, u64)
    sig("ho_inverse_mod", u64, u64, u64)
    sig("ho_multiply_mod", u64, u64, u64, u64)
    sig("ho_multiply_mod_precon", u64, u64, u64, u64, u64)
    sig("ho_multiply_mod_lazy64", u64, u64, u64, u64, u64)
    sig("ho_add_uint_mod", u64, u64, u64, u64)
    sig("ho_sub_uint_mod", u64, u64, u64, u64)
    sig("ho_pow_mod", u64, u64, u64, u64)
    sig("ho_is_primitive_root", C.c_int, u64, u64, u64)
    sig("ho_minimal_primitive_root", u64, u64, u64)
    sig("ho_reverse_bits", u64, u64, u64)
    sig("ho_is_prime", C.c_int, u64)
    sig("ho_generate_primes", C.c_size_t, p64, C.c_size_t, C.c_size_t, C.c_int,
        C.c_size_t)
    sig("ho_ntt_tables", None, u64, u64, u64, p64, p64, p64, p64)
    sig("ho_ntt_forward_radix2", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_inverse_radix2", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_forward_radix4", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_inverse_radix4", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_forward_reference", None, p64, u64, u64, p64)
    sig("ho_ntt_inverse_reference", None, p64, u64, u64, p64)
    sig("ho_eltwise_add_mod", None, p64, p64, p64, u64, u64)
    sig("ho_eltwise_add_mod_scalar", None, p64, p64, u64, u64, u64)
    sig("ho_eltwise_sub_mod", None, p64, p64, p64, u64, u64)
    sig("ho_eltwise_sub_mod_scalar", None, p64, p64, u64, u64, u64)
    sig("ho_eltwise_mult_mod", None, p64, p64, p64, u64, u64, u64)
    sig("ho_eltwise_fma_mod", None, p64, p64, u64, p64, u64, u64, u64)
    sig("ho_eltwise_reduce_mod", None, p64, p64, u64, u64, u64, u64)
    sig("ho_dyadic_multiply", None, p64, p64, p64, u64, p64, u64)
    sig("ho_eltwise_cmp_add", None, p64, p64, u64, C.c_int, u64, u64)
    sig("ho_eltwise_cmp_sub_mod", None, p64, p64, u64, u64, C.c_int, u64, u64)
    sig("ho_ntt_create", C.c_void_p, u64, u64, u64)
    sig("ho_ntt_destroy", None, C.c_void_p)
    sig("ho_ntt_forward_batch", None, C.c_void_p, p64, p64, u64, u64, u64)
    sig("ho_ntt_inverse_batch", None, C.c_void_p, p64, p64, u64, u64, u64)
    sig("ho_key_switch", None, p64, p64, u64, u64, u64, u64, u64, p64, C.POINTER(p64), p64)
    sig("ho_has_avx512", C.c_int)
    sig("ho_ntt_forward_radix2_avx512", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_inverse_radix2_avx512", None, p64, p64, u64, u64, p64, p64, u64, u64)
    sig("ho_ntt_forward_batch_avx512", None, C.c_void_p, p64, p64, u64, u64, u64)
    sig("ho_ntt_inverse_batch_avx512", None, C.c_void_p, p64, p64, u64, u64, u64)
    sig("ho_fill_splitmix", None, p64, u64, u64, u64)
    return lib


lib = _load()


def _p(a):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(p64)


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


# --- scalar number theory -------------------------------------------------
mul_hi64 = lib.ho_mul_hi64
msb = lib.ho_msb
multiply_factor = lib.ho_multiply_factor
inverse_mod = lib.ho_inverse_mod
multiply_mod = lib.ho_multiply_mod
multiply_mod_precon = lib.ho_multiply_mod_precon
multiply_mod_lazy64 = lib.ho_multiply_mod_lazy64
add_uint_mod = lib.ho_add_uint_mod
sub_uint_mod = lib.ho_sub_uint_mod
pow_mod = lib.ho_pow_mod
minimal_primitive_root = lib.ho_minimal_primitive_root
reverse_bits = lib.ho_reverse_bits
divide_u128_u64_lo = lib.ho_divide_u128_u64_lo


def is_primitive_root(root, degree, modulus):
    return bool(lib.ho_is_primitive_root(root, degree, modulus))


def is_prime(n):
    return bool(lib.ho_is_prime(n))


def generate_primes(num_primes, bit_size, prefer_small_primes, ntt_size=1):
    out = np.zeros(num_primes, dtype=np.uint64)
    found = lib.ho_generate_primes(_p(out), num_primes, bit_size,
                                   int(bool(prefer_small_primes)), ntt_size)
    if found != num_primes:
        raise RuntimeError("Failed to find enough primes")
    return [int(x) for x in out]


def fill_splitmix(n, seed, bound):
    out = np.empty(n, dtype=np.uint64)
    lib.ho_fill_splitmix(_p(out), n, seed, bound)
    return out


# --- NTT --------------------------------------------------------------------
class NTT:
    """Mirror of intel::hexl::NTT on the oracle (native radix-2 path)."""

    def __init__(self, degree, q, root_of_unity=0):
        self.n = int(degree)
        self.q = int(q)
        self.w = int(root_of_unity) if root_of_unity else int(
            lib.ho_minimal_primitive_root(2 * self.n, self.q))
        t = np.zeros((4, self.n), dtype=np.uint64)
        lib.ho_ntt_tables(self.n, self.q, self.w, _p(t[0]), _p(t[1]), _p(t[2]),
                          _p(t[3]))
        self.root_pows, self.precon_root_pows = t[0], t[1]
        self.inv_root_pows, self.precon_inv_root_pows = t[2], t[3]

    def forward(self, operand, in_mf=1, out_mf=1):
        """operand: (..., n) uint64; returns a new array (out-of-place)."""
        x = _arr(operand)
        flat = x.reshape(-1, self.n)
        out = np.empty_like(flat)
        for b in range(flat.shape[0]):
            lib.ho_ntt_forward_radix2(_p(out[b]), _p(flat[b]), self.n, self.q,
                                      _p(self.root_pows),
                                      _p(self.precon_root_pows), in_mf, out_mf)
        return out.reshape(x.shape)

    def inverse(self, operand, in_mf=1, out_mf=1):
        x = _arr(operand)
        flat = x.reshape(-1, self.n)
        out = np.empty_like(flat)
        for b in range(flat.shape[0]):
            lib.ho_ntt_inverse_radix2(_p(out[b]), _p(flat[b]), self.n, self.q,
                                      _p(self.inv_root_pows),
                                      _p(self.precon_inv_root_pows), in_mf,
                                      out_mf)
        return out.reshape(x.shape)

    def forward_radix4(self, operand, in_mf=1, out_mf=1, inplace=False):
        """The reference's radix-4 native transform (ntt-radix-4.cpp:17-400), one polynomial."""
        x = _arr(operand).copy()
        out = x if inplace else np.empty_like(x)
        lib.ho_ntt_forward_radix4(_p(out), _p(x), self.n, self.q, _p(self.root_pows),
                                  _p(self.precon_root_pows), in_mf, out_mf)
        return out

    def inverse_radix4(self, operand, in_mf=1, out_mf=1, inplace=False):
        """ntt-radix-4.cpp:402-700"""
        x = _arr(operand).copy()
        out = x if inplace else np.empty_like(x)
        lib.ho_ntt_inverse_radix4(_p(out), _p(x), self.n, self.q, _p(self.inv_root_pows),
                                  _p(self.precon_inv_root_pows), in_mf, out_mf)
        return out

    def forward_inplace(self, buf, in_mf=1, out_mf=1):
        assert buf.dtype == np.uint64 and buf.size == self.n
        lib.ho_ntt_forward_radix2(_p(buf), _p(buf), self.n, self.q,
                                  _p(self.root_pows),
                                  _p(self.precon_root_pows), in_mf, out_mf)

    def inverse_inplace(self, buf, in_mf=1, out_mf=1):
        assert buf.dtype == np.uint64 and buf.size == self.n
        lib.ho_ntt_inverse_radix2(_p(buf), _p(buf), self.n, self.q,
                                  _p(self.inv_root_pows),
                                  _p(self.precon_inv_root_pows), in_mf, out_mf)

    def forward_reference(self, operand):
        x = _arr(operand).copy()
        lib.ho_ntt_forward_reference(_p(x), self.n, self.q, _p(self.root_pows))
        return x

    def inverse_reference(self, operand):
        x = _arr(operand).copy()
        lib.ho_ntt_inverse_reference(_p(x), self.n, self.q,
                                     _p(self.inv_root_pows))
        return x


# --- element-wise -------------------------------------------------------------
def eltwise_add_mod(a, b, q):
    a = _arr(a)
    out = np.empty_like(a)
    if np.isscalar(b) or isinstance(b, int):
        lib.ho_eltwise_add_mod_scalar(_p(out), _p(a), int(b), a.size, q)
    else:
        b = _arr(b)
        lib.ho_eltwise_add_mod(_p(out), _p(a), _p(b), a.size, q)
    return out


def eltwise_sub_mod(a, b, q):
    a = _arr(a)
    out = np.empty_like(a)
    if np.isscalar(b) or isinstance(b, int):
        lib.ho_eltwise_sub_mod_scalar(_p(out), _p(a), int(b), a.size, q)
    else:
        b = _arr(b)
        lib.ho_eltwise_sub_mod(_p(out), _p(a), _p(b), a.size, q)
    return out


def eltwise_mult_mod(a, b, q, in_mf=1):
    a, b = _arr(a), _arr(b)
    out = np.empty_like(a)
    lib.ho_eltwise_mult_mod(_p(out), _p(a), _p(b), a.size, q, in_mf)
    return out


def eltwise_fma_mod(a, s, c, q, in_mf=1):
    a = _arr(a)
    out = np.empty_like(a)
    cp = None
    if c is not None:
        c = _arr(c)
        cp = _p(c)
    lib.ho_eltwise_fma_mod(_p(out), _p(a), int(s), cp, a.size, q, in_mf)
    return out


# CMPINT (hexl/include/hexl/util/util.hpp:16-25)
CMPINT = {"EQ": 0, "LT": 1, "LE": 2, "FALSE": 3, "NE": 4, "NLT": 5, "NLE": 6, "TRUE": 7}


def eltwise_cmp_add(a, cmp, bound, diff):
    a = _arr(a)
    out = np.empty_like(a)
    lib.ho_eltwise_cmp_add(_p(out), _p(a), a.size, int(cmp), int(bound), int(diff))
    return out


def eltwise_cmp_sub_mod(a, modulus, cmp, bound, diff):
    a = _arr(a)
    out = np.empty_like(a)
    lib.ho_eltwise_cmp_sub_mod(_p(out), _p(a), a.size, int(modulus), int(cmp), int(bound),
                               int(diff))
    return out


def dyadic_multiply(op1, op2, n, moduli, result=None):
    """DyadicMultiply; op1/op2 hold 2 polynomials of n * len(moduli) words, the result 3.
    `result` (an array of 3 * n * len(moduli) words) may be op1 / op2 themselves to
    exercise the in-place forms of the reference's tests."""
    op1, op2, moduli = _arr(op1), _arr(op2), _arr(moduli)
    if result is None:
        result = np.zeros(3 * n * moduli.size, dtype=np.uint64)
    lib.ho_dyadic_multiply(_p(result), _p(op1), _p(op2), n, _p(moduli), moduli.size)
    return result


def key_switch(result, t_target, n, decomp, key_mod, rns_mod, key_comp, moduli, keys,
               modswitch_factors):
    """KeySwitch; `result` (key_comp * decomp * n words) is accumulated into and returned;
    `keys` is a list of `decomp` arrays of key_comp * key_mod * n words."""
    result, t_target = _arr(result).copy(), _arr(t_target)
    moduli, msf = _arr(moduli), _arr(modswitch_factors)
    keys = [_arr(k) for k in keys]
    kp = (p64 * len(keys))(*[C.cast(_p(k), p64) for k in keys])
    lib.ho_key_switch(_p(result), _p(t_target), n, decomp, key_mod, rns_mod, key_comp,
                      _p(moduli), kp, _p(msf))
    return result


def eltwise_reduce_mod(a, q, in_mf, out_mf):
    a = _arr(a)
    out = np.empty_like(a)
    lib.ho_eltwise_reduce_mod(_p(out), _p(a), a.size, q, in_mf, out_mf)
    return out
