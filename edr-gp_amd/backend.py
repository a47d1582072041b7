"""Python handle on one native context (one HIP device, one stream).

Thin: every method is one C-ABI call (include/edrgp_hip.h).  Arrays cross the
boundary as row-major float64; training data is copied to HBM once by
``set_data`` and stays resident across evaluations.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from ._lib import KERN_CODES, as_f64, check, dptr


class HipGP:
    def __init__(self, device=0):
        self._lib = _lib.load()
        self._ctx = ctypes.c_void_p()
        check(self._lib.edrgp_ctx_create(int(device), ctypes.byref(self._ctx)))
        self.device = int(device)
        self.N = self.D = self.M = 0
        self.nparts = 0

    # -- lifetime ----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self._lib.edrgp_ctx_destroy(self._ctx)
            self._ctx = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        check(self._lib.edrgp_sync(self._ctx))

    def set_option(self, key, value):
        check(self._lib.edrgp_set_option(self._ctx, key.encode(), float(value)))

    # -- state -------------------------------------------------------------
    def set_data(self, X, y):
        X = as_f64(X)
        y = as_f64(np.ravel(y))
        if X.ndim != 2 or y.shape[0] != X.shape[0]:
            raise ValueError("X must be (N, D) and y (N,)")
        check(self._lib.edrgp_set_data(self._ctx, dptr(X), dptr(y), X.shape[0], X.shape[1]))
        self.N, self.D = X.shape

    def set_data_projected(self, W, Xbase=None, y=None):
        """X = Xbase @ W on the device; Xbase / y are uploaded when given and stay resident otherwise."""
        W = as_f64(W)
        if Xbase is not None:
            Xbase = as_f64(Xbase)
            y = as_f64(np.ravel(y))
            if Xbase.ndim != 2 or y.shape[0] != Xbase.shape[0] or W.shape[0] != Xbase.shape[1]:
                raise ValueError("Xbase must be (N, Dbase), y (N,), W (Dbase, d)")
            self._base_shape = Xbase.shape
        elif getattr(self, "_base_shape", None) is None or W.shape[0] != self._base_shape[1]:
            raise ValueError("no resident base rows matching W")
        n, dbase = self._base_shape
        check(self._lib.edrgp_set_data_projected(self._ctx, dptr(Xbase), dptr(y), n, dbase, dptr(W), W.shape[1]))
        self.N, self.D = n, W.shape[1]

    def set_inducing(self, Z):
        Z = as_f64(Z)
        if Z.ndim != 2 or Z.shape[1] != self.D:
            raise ValueError("Z must be (M, %d)" % self.D)
        check(self._lib.edrgp_set_inducing(self._ctx, dptr(Z), Z.shape[0]))
        self.M = Z.shape[0]

    def set_kernel(self, kinds, variances, lengthscales, noise_var):
        """kinds: list of names; lengthscales: (nparts, D) expanded per dimension."""
        codes = np.ascontiguousarray([KERN_CODES[k] for k in kinds], dtype=np.int32)
        var = as_f64(variances).ravel()
        ls = as_f64(lengthscales).reshape(len(kinds), self.D)
        check(self._lib.edrgp_set_kernel(self._ctx, len(kinds), codes.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                         dptr(var), dptr(ls), float(noise_var)))
        self.nparts = len(kinds)
        self._kernel = (list(kinds), var.copy(), ls.copy())

    def kernel_matrix(self, kinds, variances, lengthscales, X, X2=None):
        """K(X, X2) (``X2=None``: K(X, X)) of the kernel given by value; stateless -- nothing resident on the
        device (data, Z, hyper-parameters, posterior) is touched."""
        X = as_f64(X)
        D = X.shape[1]
        codes = np.ascontiguousarray([KERN_CODES[k] for k in kinds], dtype=np.int32)
        var = as_f64(variances).ravel()
        ls = as_f64(lengthscales).reshape(len(kinds), D)
        X2c = None if X2 is None else as_f64(X2)
        n2 = X.shape[0] if X2c is None else X2c.shape[0]
        out = np.empty((X.shape[0], n2))
        check(self._lib.edrgp_kernel_matrix(self._ctx, len(kinds), codes.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                                            dptr(var), dptr(ls), dptr(X), X.shape[0], dptr(X2c), n2, D, dptr(out)))
        return out

    # -- evaluations -------------------------------------------------------
    def sparse_lml(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_sparse_lml(self._ctx, ctypes.byref(out)))
        return out.value

    def sparse_lml_grad(self):
        lml = ctypes.c_double(0.0)
        dnoise = ctypes.c_double(0.0)
        dvar = np.zeros(self.nparts)
        dlen = np.zeros((self.nparts, self.D))
        dZ = np.zeros((self.M, self.D))
        check(self._lib.edrgp_sparse_lml_grad(self._ctx, ctypes.byref(lml), dptr(dvar), dptr(dlen),
                                              ctypes.byref(dnoise), dptr(dZ)))
        return lml.value, dvar, dlen, dnoise.value, dZ

    def sparse_posterior(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_sparse_posterior(self._ctx, ctypes.byref(out)))
        return out.value

    def dense_lml(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_dense_lml_grad(self._ctx, ctypes.byref(out), None, None, None))
        return out.value

    def dense_lml_grad(self):
        lml = ctypes.c_double(0.0)
        dnoise = ctypes.c_double(0.0)
        dvar = np.zeros(self.nparts)
        dlen = np.zeros((self.nparts, self.D))
        check(self._lib.edrgp_dense_lml_grad(self._ctx, ctypes.byref(lml), dptr(dvar), dptr(dlen),
                                             ctypes.byref(dnoise)))
        return lml.value, dvar, dlen, dnoise.value

    def dense_posterior(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_dense_posterior(self._ctx, ctypes.byref(out)))
        return out.value

    # -- prediction (latent, normalised space) -----------------------------
    def predict(self, Xs, want_var=True):
        Xs = as_f64(Xs)
        mean = np.empty(Xs.shape[0])
        var = np.empty(Xs.shape[0]) if want_var else None
        check(self._lib.edrgp_predict(self._ctx, dptr(Xs), Xs.shape[0], dptr(mean), dptr(var)))
        return mean, var

    def predict_grad(self, Xs):
        Xs = as_f64(Xs)
        grad = np.empty((Xs.shape[0], self.D))
        check(self._lib.edrgp_predict_grad(self._ctx, dptr(Xs), Xs.shape[0], dptr(grad)))
        return grad

    def predict_train(self, want_mean=True, want_grad=True):
        mean = np.empty(self.N) if want_mean else None
        grad = np.empty((self.N, self.D)) if want_grad else None
        check(self._lib.edrgp_predict_train(self._ctx, dptr(mean), dptr(grad)))
        return mean, grad

    def predict_train_gram(self, want_mean=False):
        """J^T J (D x D) and the column sums of J = d mean / d x over the training rows, reduced on the device."""
        mean = np.empty(self.N) if want_mean else None
        gram = np.empty((self.D, self.D))
        gsum = np.empty(self.D)
        check(self._lib.edrgp_predict_train_gram(self._ctx, dptr(mean), dptr(gram), dptr(gsum)))
        return mean, gram, gsum

    # -- classification (EP) -----------------------------------------------------
    def ep_begin(self, y01):
        y = as_f64(np.ravel(y01))
        if y.shape[0] != self.N:
            raise ValueError("labels must have one entry per resident row")
        check(self._lib.edrgp_ep_begin(self._ctx, dptr(y)))

    def ep_sweep(self, order, eta=1.0, delta=1.0):
        order = np.ascontiguousarray(order, dtype=np.int32)
        dt, dv = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(self._lib.edrgp_ep_sweep(self._ctx, order.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), float(eta), float(delta),
                                       ctypes.byref(dt), ctypes.byref(dv)))
        return dt.value, dv.value

    def ep_finish(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_ep_finish(self._ctx, ctypes.byref(out)))
        return out.value

    def ep_get_sites(self):
        tau, v = np.empty(self.N), np.empty(self.N)
        check(self._lib.edrgp_ep_get_sites(self._ctx, dptr(tau), dptr(v)))
        return tau, v

    def ep_set_sites(self, tau, v, log_Z_tilde):
        check(self._lib.edrgp_ep_set_sites(self._ctx, dptr(as_f64(tau)), dptr(as_f64(v)), float(log_Z_tilde)))

    def ep_lml(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_ep_lml_grad(self._ctx, ctypes.byref(out), None, None))
        return out.value

    def ep_lml_grad(self):
        lml = ctypes.c_double(0.0)
        dvar = np.zeros(self.nparts)
        dlen = np.zeros((self.nparts, self.D))
        check(self._lib.edrgp_ep_lml_grad(self._ctx, ctypes.byref(lml), dptr(dvar), dptr(dlen)))
        return lml.value, dvar, dlen

    def ep_posterior(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_ep_posterior(self._ctx, ctypes.byref(out)))
        return out.value

    # -- sparse EP classifier (EPDTC); ep_finish / ep_get_sites / ep_set_sites are shared with the dense one
    def epdtc_begin(self, y01):
        y = as_f64(np.ravel(y01))
        if y.shape[0] != self.N:
            raise ValueError("labels must have one entry per resident row")
        check(self._lib.edrgp_epdtc_begin(self._ctx, dptr(y)))

    def epdtc_sweep(self, order, eta=1.0, delta=1.0):
        order = np.ascontiguousarray(order, dtype=np.int32)
        dt, dv = ctypes.c_double(0.0), ctypes.c_double(0.0)
        check(self._lib.edrgp_epdtc_sweep(self._ctx, order.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), float(eta),
                                          float(delta), ctypes.byref(dt), ctypes.byref(dv)))
        return dt.value, dv.value

    def epdtc_lml(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_epdtc_lml_grad(self._ctx, ctypes.byref(out), None, None, None))
        return out.value

    def epdtc_lml_grad(self):
        lml = ctypes.c_double(0.0)
        dvar = np.zeros(self.nparts)
        dlen = np.zeros((self.nparts, self.D))
        dZ = np.zeros((self.M, self.D))
        check(self._lib.edrgp_epdtc_lml_grad(self._ctx, ctypes.byref(lml), dptr(dvar), dptr(dlen), dptr(dZ)))
        return lml.value, dvar, dlen, dZ

    def epdtc_posterior(self):
        out = ctypes.c_double(0.0)
        check(self._lib.edrgp_epdtc_posterior(self._ctx, ctypes.byref(out)))
        return out.value

    def sweep_scales(self, lengthscales, variance=1.0, noise_var=0.01, kind="RBF", want_gram=True):
        """Bound (and J^T J / column sums of the mean Jacobian at the training rows) for a grid of isotropic
        length-scales of the resident rows and inducing inputs, as ONE batch on the device.
        Returns (bounds[G], grams[G, D, D] or None, gsums[G, D] or None, modes[G]); mode -1 = not evaluated
        (needs jitter: take the per-unit path for that scale)."""
        ls = as_f64(lengthscales).ravel()
        G = ls.size
        bounds = np.empty(G)
        grams = np.empty((G, self.D, self.D)) if want_gram else None
        gsums = np.empty((G, self.D)) if want_gram else None
        modes = np.zeros(G, dtype=np.int32)
        check(self._lib.edrgp_sweep_scales(self._ctx, G, dptr(ls), float(variance), float(noise_var), KERN_CODES[kind],
                                           dptr(bounds), dptr(grams), dptr(gsums),
                                           modes.ctypes.data_as(ctypes.POINTER(ctypes.c_int))))
        return bounds, grams, gsums, modes

    # -- parity / profiling accessors --------------------------------------
    def get(self, name, shape):
        buf = np.empty(shape)
        check(self._lib.edrgp_get(self._ctx, _lib.GET[name], dptr(buf), buf.size))
        return buf

    def profile_reset(self):
        check(self._lib.edrgp_profile_reset(self._ctx))

    def profile_get(self, name):
        ms = ctypes.c_double(0.0)
        n = ctypes.c_int64(0)
        check(self._lib.edrgp_profile_get(self._ctx, _lib.PROF[name], ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def calib(self):
        tf = ctypes.c_double(0.0)
        bw = ctypes.c_double(0.0)
        check(self._lib.edrgp_calib(self._ctx, ctypes.byref(tf), ctypes.byref(bw)))
        return tf.value, bw.value

    # -- multi-GPU ----------------------------------------------------------
    @staticmethod
    def comm_unique_id():
        buf = ctypes.create_string_buffer(128)
        check(_lib.load().edrgp_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        check(self._lib.edrgp_comm_init(self._ctx, int(nranks), int(rank), buf))

    def comm_init_host(self, nranks, rank, allreduce):
        """Row sharding over a collective the launcher provides: ``allreduce(a)`` must sum the float64 numpy
        array ``a`` over all ranks IN PLACE (e.g. ``dist.all_reduce(torch.from_numpy(a))`` on a gloo group)."""
        def _cb(_user, buf, count):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(count,)))
                return 0
            except Exception:           # noqa: BLE001  (an exception must not unwind through the C frame)
                import traceback
                traceback.print_exc()
                return 1
        self._host_ar = _lib.ALLREDUCE_FN(_cb)          # keep the trampoline alive as long as it is attached
        check(self._lib.edrgp_comm_init_host(self._ctx, int(nranks), int(rank),
                                             ctypes.cast(self._host_ar, ctypes.c_void_p), None))

    def comm_destroy(self):
        check(self._lib.edrgp_comm_destroy(self._ctx))
        self._host_ar = None

    # -- low-level test hooks -----------------------------------------------
    def dbg_gemm(self, A, B, transA=False, transB=False, alpha=1.0, beta=0.0, C=None):
        A = as_f64(A)
        B = as_f64(B)
        M = A.shape[1] if transA else A.shape[0]
        K = A.shape[0] if transA else A.shape[1]
        N = B.shape[0] if transB else B.shape[1]
        out = np.zeros((M, N)) if C is None else as_f64(C).copy()
        check(self._lib.edrgp_dbg_gemm(self._ctx, int(transA), int(transB), M, N, K, float(alpha), dptr(A),
                                       A.shape[1], dptr(B), B.shape[1], float(beta), dptr(out), N))
        return out

    def dbg_gemm32(self, A, B, transA=False, transB=False):
        A = np.ascontiguousarray(A, dtype=np.float32)
        B = np.ascontiguousarray(B, dtype=np.float32)
        M = A.shape[1] if transA else A.shape[0]
        K = A.shape[0] if transA else A.shape[1]
        N = B.shape[0] if transB else B.shape[1]
        out = np.zeros((M, N), dtype=np.float32)
        fp = ctypes.POINTER(ctypes.c_float)
        check(self._lib.edrgp_dbg_gemm32(self._ctx, int(transA), int(transB), M, N, K, A.ctypes.data_as(fp),
                                         A.shape[1], B.ctypes.data_as(fp), B.shape[1], out.ctypes.data_as(fp), N))
        return out

    def dbg_syrk32(self, K):
        K = np.ascontiguousarray(K, dtype=np.float32)
        out = np.empty((K.shape[1], K.shape[1]))
        check(self._lib.edrgp_dbg_syrk32(self._ctx, K.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), K.shape[0],
                                         K.shape[1], dptr(out)))
        return out

    def dbg_potrf(self, A):
        L = as_f64(A).copy()
        info = ctypes.c_int(0)
        check(self._lib.edrgp_dbg_potrf(self._ctx, dptr(L), L.shape[0], ctypes.byref(info)))
        return L, info.value

    def dbg_potrf_inv(self, A):
        L = as_f64(A).copy()
        Li = np.empty_like(L)
        info = ctypes.c_int(0)
        check(self._lib.edrgp_dbg_potrf_inv(self._ctx, dptr(L), L.shape[0], ctypes.byref(info), dptr(Li)))
        return L, Li, info.value

    def dbg_potrf_time(self, A, reps=5):
        """Average time (ms) of the production factorisation + inverse of the factor alone on the SPD matrix A."""
        A = as_f64(A)
        ms = ctypes.c_double(0.0)
        check(self._lib.edrgp_dbg_potrf_time(self._ctx, dptr(A), A.shape[0], reps, ctypes.byref(ms)))
        return ms.value

    def dbg_trtri(self, L):
        L = as_f64(L)
        out = np.empty_like(L)
        check(self._lib.edrgp_dbg_trtri(self._ctx, dptr(L), dptr(out), L.shape[0]))
        return out

    def dbg_syrk(self, K):
        K = as_f64(K)
        out = np.empty((K.shape[1], K.shape[1]))
        check(self._lib.edrgp_dbg_syrk(self._ctx, dptr(K), K.shape[0], K.shape[1], dptr(out)))
        return out

    def dbg_syrk_time(self, N, M, splits=0, reps=3, want_phi=False):
        """Average time (ms) of the production Phi = K^T K kernel alone on a device-generated K."""
        ms = ctypes.c_double(0.0)
        phi = np.empty((M, M)) if want_phi else None
        check(self._lib.edrgp_dbg_syrk_time(self._ctx, N, M, splits, reps, ctypes.byref(ms), dptr(phi)))
        return ms.value, phi

    def dbg_kernel_matrix(self, X, X2=None):
        """K(X, X2) with the hyper-parameters of the last ``set_kernel`` (test convenience)."""
        kinds, var, ls = self._kernel
        return self.kernel_matrix(kinds, var, ls, X, X2)

    def dbg_mfma_layout(self, A16x4, B4x16):
        A = as_f64(A16x4)
        B = as_f64(B4x16)
        out = np.empty((16, 16))
        check(self._lib.edrgp_dbg_mfma_layout(self._ctx, dptr(A), dptr(B), dptr(out)))
        return out
