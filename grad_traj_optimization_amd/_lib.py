"""ctypes binding of include/gtop.h (the C-ABI of libgtop_hip.so)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GTOP_HIP_LIB selects another build of the same C-ABI (kernel tuning variants)
_SO = os.environ.get("GTOP_HIP_LIB") or os.path.join(_HERE, "libgtop_hip.so")

GTOP_F64, GTOP_F32 = 0, 1
_STATUS = {0: "GTOP_OK", 1: "GTOP_ERR_INVALID", 2: "GTOP_ERR_HIP", 3: "GTOP_ERR_NO_DEVICE",
           4: "GTOP_ERR_STATE"}


class GtopError(RuntimeError):
    def __init__(self, code, msg=""):
        self.code = code
        super().__init__(f"{_STATUS.get(code, code)}: {msg}")


class GtopParams(C.Structure):
    """gtop_params — the ROS parameters the callback reads
    (src/grad_traj_optimizer.cpp:5-32 of the reference) + step + enable_dyn."""
    _fields_ = [
        ("ws", C.c_double), ("wc", C.c_double),
        ("alpha", C.c_double), ("r", C.c_double), ("d0", C.c_double),
        ("alpha_v", C.c_double), ("r_v", C.c_double), ("v0", C.c_double),
        ("alpha_a", C.c_double), ("r_a", C.c_double), ("a0", C.c_double),
        ("step", C.c_int32), ("enable_dyn", C.c_int32),
    ]


class GtopStop(C.Structure):
    """gtop_stop — evaluation cap + NLopt's ftol_rel / xtol_rel / maxtime."""
    _fields_ = [("max_evals", C.c_int32), ("ftol_rel", C.c_double), ("xtol_rel", C.c_double), ("maxtime", C.c_double)]


# launch/opti_node.launch:3-28 of the reference — the only parameter set whose
# names match what the ctor reads.
OPTI_NODE_PARAMS = dict(ws=1.0, wc=5.0, alpha=10.0, r=0.5, d0=0.8,
                        alpha_v=0.0, r_v=1.5, v0=2.5, alpha_a=0.0, r_a=1.5, a0=3.5,
                        step=2, enable_dyn=0)

_lib = None


def library_path():
    return _SO


def _preload_torch_hip_runtime():
    """One process must hold ONE HIP runtime.  The PyTorch-ROCm wheel bundles
    its own libamdhip64.so (SONAME libamdhip64.so.7, the name libgtop_hip.so
    needs): if libgtop_hip.so were loaded first it would pull /opt/rocm's copy,
    a later `import torch` would add the bundled one, and whichever initialises
    second sees "no ROCm-capable device".  So when torch is installed, map its
    copy first (by path, without importing torch); the loader then resolves our
    NEEDED entry to it by SONAME, and torch finds it already loaded."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load_library():
    """Load libgtop_hip.so.  Raises (never falls back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise ImportError(
            f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C grad_traj_optimization_amd/csrc` — there is no CPU fallback")
    _preload_torch_hip_runtime()
    L = C.CDLL(_SO)
    vp, dp, ip = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)
    sig = {
        "gtop_abi_version": (C.c_int, []),
        "gtop_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "gtop_destroy": (C.c_int, [vp]),
        "gtop_last_error": (C.c_char_p, [vp]),
        "gtop_set_params": (C.c_int, [vp, C.POINTER(GtopParams)]),
        "gtop_set_sdf": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double]),
        "gtop_set_sdf_device": (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double]),
        "gtop_init_sdf_map": (C.c_int, [vp, dp, dp, C.c_double]),
        "gtop_update_sdf_map": (C.c_int, [vp, dp, C.c_int]),
        "gtop_update_sdf_map_device": (C.c_int, [vp, vp, C.c_int, vp]),
        "gtop_get_sdf": (C.c_int, [vp, dp, ip]),
        "gtop_set_problem": (C.c_int, [vp, C.c_int, C.c_int, dp, C.c_int, dp]),
        "gtop_eval_batch": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "gtop_cost_nlopt": (C.c_double, [C.c_uint, dp, dp, vp]),
        "gtop_eval_device": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, vp]),
        "gtop_set_paths": (C.c_int, [vp, C.c_int, C.c_int, dp, C.c_double, C.c_double, dp]),
        "gtop_setup_paths_device": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_double, C.c_double, vp, vp, vp, vp]),
        "gtop_get_problem": (C.c_int, [vp, dp, dp]),
        "gtop_coefficients_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp]),
        "gtop_eval_trajectories_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_double, vp, vp]),
        "gtop_trajectory_stats": (C.c_int, [vp, C.c_int, dp, C.c_double, dp, dp]),
        "gtop_set_moving_boxes": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "gtop_edt_query_device": (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp]),
        "gtop_edt_query": (C.c_int, [vp, C.c_int, dp, dp, dp, dp]),
        "gtop_edt_coarse_query_device": (C.c_int, [vp, C.c_int, vp, vp, vp, vp]),
        "gtop_edt_coarse_query": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "gtop_sample_trajectories_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, C.c_double, vp, vp, C.c_int, vp]),
        "gtop_trajectory_samples": (C.c_int, [vp, C.c_int, dp, C.c_double, dp, dp, dp, C.c_int]),
        "gtop_default_bounds": (C.c_int, [C.c_int, C.c_int, dp, C.c_double, C.c_double, C.c_double, dp, dp]),
        "gtop_optimize_batch": (C.c_int, [vp, C.c_int, dp, dp, dp, C.c_int, dp]),
        "gtop_optimize_device": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp]),
        "gtop_optimize_batch_ex": (C.c_int, [vp, C.c_int, dp, dp, dp, C.POINTER(GtopStop), dp, C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32)]),
        "gtop_optimize_device_ex": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp, vp, C.POINTER(GtopStop), vp,
                                              vp, vp, vp]),
        "gtop_get_stats": (C.c_int, [vp, C.POINTER(C.c_int64), dp]),
        "gtop_reset_stats": (C.c_int, [vp]),
        "gtop_get_cost_curve": (C.c_int, [vp, dp, dp, C.c_int, ip]),
        "gtop_clear_cost_curve": (C.c_int, [vp]),
        "gtop_set_launch_geometry": (C.c_int, [vp, C.c_int, C.c_int]),
        "gtop_update_sdf_map_window": (C.c_int, [vp, dp, dp, dp, C.c_int]),
        "gtop_update_sdf_map_window_device": (C.c_int, [vp, dp, dp, vp, C.c_int, vp]),
        "gtop_set_field_precisions": (C.c_int, [vp, C.c_int]),
        "gtop_device_clock_stamp": (C.c_int, [vp, vp, vp]),
        "gtop_push_rows": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(C.c_void_p), C.c_int, vp, vp]),
        "gtop_shared_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(C.c_void_p), C.c_char_p]),
        "gtop_shared_open": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
        "gtop_shared_close": (C.c_int, [vp, vp]),
        "gtop_shared_free": (C.c_int, [vp, vp]),
        "gtop_device_clock_hz": (C.c_int, [vp, dp]),
        "gtop_rendezvous_create": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int]),
        "gtop_rendezvous_destroy": (C.c_int, [vp]),
        "gtop_rendezvous_get_slot": (vp, [vp, C.c_int]),
        "gtop_cost_nlopt_shared": (C.c_double, [C.c_uint, dp, dp, vp]),
        "gtop_rendezvous_leave": (C.c_int, [vp]),
        "gtop_rendezvous_set_timeout": (C.c_int, [vp, C.c_double]),
        "gtop_rendezvous_abort": (C.c_int, [vp]),
        "gtop_rendezvous_stats": (C.c_int, [vp, C.POINTER(C.c_int64), dp, C.POINTER(C.c_int64)]),
        "gtop_set_optimizer_fusion": (C.c_int, [vp, C.c_int]),
        "gtop_set_optimizer_precision": (C.c_int, [vp, C.c_int]),
        "gtop_group_create": (C.c_int, [C.POINTER(vp), ip, C.c_int]),
        "gtop_group_destroy": (C.c_int, [vp]),
        "gtop_group_size": (C.c_int, [vp]),
        "gtop_group_context": (vp, [vp, C.c_int]),
        "gtop_group_last_error": (C.c_char_p, [vp]),
        "gtop_group_gather_backend": (C.c_char_p, [vp]),
        "gtop_group_gather_note": (C.c_char_p, [vp]),
        "gtop_group_update_sdf_map_window": (C.c_int, [vp, dp, dp, dp, C.c_int]),
        "gtop_group_set_params": (C.c_int, [vp, C.POINTER(GtopParams)]),
        "gtop_group_init_sdf_map": (C.c_int, [vp, dp, dp, C.c_double]),
        "gtop_group_update_sdf_map": (C.c_int, [vp, dp, C.c_int]),
        "gtop_group_set_sdf": (C.c_int, [vp, dp, C.c_int, C.c_int, C.c_int, dp, dp, C.c_double]),
        "gtop_group_set_problem": (C.c_int, [vp, C.c_int, C.c_int, dp, C.c_int, dp]),
        "gtop_group_shard": (C.c_int, [vp, C.c_int, ip, ip]),
        "gtop_group_eval_batch": (C.c_int, [vp, C.c_int, dp, dp, dp]),
        "gtop_group_upload_x": (C.c_int, [vp, C.c_int, dp]),
        "gtop_group_eval_resident": (C.c_int, [vp, C.c_int, C.c_int]),
        "gtop_group_synchronize": (C.c_int, [vp]),
        "gtop_group_read_gathered": (C.c_int, [vp, C.c_int, dp, dp]),
        "gtop_group_device_buffers": (C.c_int, [vp, C.c_int] + [C.POINTER(vp)] * 6),
        "gtop_group_optimize_batch_ex": (C.c_int, [vp, C.c_int, dp, dp, dp, C.POINTER(GtopStop), dp, C.POINTER(C.c_int32),
                                                   C.POINTER(C.c_int32)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class GtopContext:
    """One gtop_ctx (one per host thread / HIP stream)."""

    def __init__(self, device=0, params=None):
        self._L = load_library()
        h = C.c_void_p()
        rc = self._L.gtop_create(C.byref(h), int(device))
        if rc != 0:
            raise GtopError(rc, self._L.gtop_last_error(None).decode()
                            + " — a gfx950 GPU is required; there is no CPU fallback")
        self._h = h
        self.device = int(device)
        self.set_params(**(params or {}))

    # -- helpers --
    def _chk(self, rc):
        if rc != 0:
            raise GtopError(rc, self._L.gtop_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.gtop_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration --
    def set_params(self, **kw):
        d = dict(OPTI_NODE_PARAMS)
        d.update(kw)
        self.params = d
        p = GtopParams(**d)
        self._chk(self._L.gtop_set_params(self._h, C.byref(p)))

    def set_sdf(self, dist, grid, origin, resolution, map_size=None):
        dist = _f64(dist).reshape(-1)
        nx, ny, nz = (int(g) for g in grid)
        assert dist.size == nx * ny * nz
        ms = _p(_f64(map_size)) if map_size is not None else None
        self._chk(self._L.gtop_set_sdf(self._h, _p(dist), nx, ny, nz, _p(_f64(origin)), ms, float(resolution)))
        self.grid = (nx, ny, nz)

    def set_sdf_device(self, tensor, grid, origin, resolution, map_size=None):
        import torch
        dtype = {torch.float64: GTOP_F64, torch.float32: GTOP_F32}[tensor.dtype]
        nx, ny, nz = (int(g) for g in grid)
        assert tensor.is_cuda and tensor.is_contiguous() and tensor.numel() == nx * ny * nz
        ms = _p(_f64(map_size)) if map_size is not None else None
        self._chk(self._L.gtop_set_sdf_device(self._h, dtype, C.c_void_p(tensor.data_ptr()), nx, ny, nz,
                                              _p(_f64(origin)), ms, float(resolution)))
        self._sdf_keepalive = tensor
        self.grid = (nx, ny, nz)

    def init_sdf_map(self, map_size, origin, resolution):
        self._chk(self._L.gtop_init_sdf_map(self._h, _p(_f64(map_size)), _p(_f64(origin)), float(resolution)))
        g = (C.c_int * 3)()
        self._chk(self._L.gtop_get_sdf(self._h, None, g))
        self.grid = tuple(g)

    def update_sdf_map(self, pts):
        pts = _f64(pts).reshape(-1, 3)
        self._chk(self._L.gtop_update_sdf_map(self._h, _p(pts), pts.shape[0]))

    def update_sdf_map_device(self, pts, stream=None):
        """pts: (N, 3) float64 CUDA tensor; asynchronous on `stream` (default: torch's current stream)."""
        import torch
        assert pts.is_cuda and pts.dtype == torch.float64 and pts.dim() == 2 and pts.shape[1] == 3
        pts = pts.contiguous()   # (a tensor made from np.argwhere's transposed view is column-major)
        st = stream if stream is not None else torch.cuda.current_stream()
        self._chk(self._L.gtop_update_sdf_map_device(self._h, C.c_void_p(pts.data_ptr()), pts.shape[0],
                                                     C.c_void_p(st.cuda_stream)))

    def update_sdf_map_window(self, min_pos, max_pos, pts):
        """The reference's local update (compare2.cpp:147-152): resetBuffer(min, max), setOccupancy per point,
        updateESDF3d over the box (gtop_update_sdf_map_window)."""
        pts = _f64(pts).reshape(-1, 3)
        mn, mx = _f64(min_pos).reshape(3), _f64(max_pos).reshape(3)
        self._chk(self._L.gtop_update_sdf_map_window(self._h, _p(mn), _p(mx), _p(pts) if pts.size else None, pts.shape[0]))

    def update_sdf_map_window_device(self, min_pos, max_pos, pts, stream=None):
        import torch
        assert pts.is_cuda and pts.dtype == torch.float64 and pts.dim() == 2 and pts.shape[1] == 3
        pts = pts.contiguous()
        mn, mx = _f64(min_pos).reshape(3), _f64(max_pos).reshape(3)
        st = stream if stream is not None else torch.cuda.current_stream()
        self._chk(self._L.gtop_update_sdf_map_window_device(self._h, _p(mn), _p(mx), C.c_void_p(pts.data_ptr()),
                                                            pts.shape[0], C.c_void_p(st.cuda_stream)))

    def get_sdf(self):
        g = (C.c_int * 3)()
        self._chk(self._L.gtop_get_sdf(self._h, None, g))
        out = np.empty(g[0] * g[1] * g[2])
        self._chk(self._L.gtop_get_sdf(self._h, _p(out), g))
        return out.reshape(tuple(g))

    def set_problem(self, T, Df):
        Df = _f64(Df)
        B = Df.size // 18
        T = _f64(T)
        if T.ndim == 2:
            m, stride = T.shape[1], T.shape[1]
            assert T.shape[0] == B
        else:
            m, stride = T.shape[0], 0
        self._chk(self._L.gtop_set_problem(self._h, B, m, _p(T), stride, _p(Df)))
        self.B, self.m = B, m

    def set_optimizer_fusion(self, mode=2):
        """2 (or True): whole optimizer loop in one launch; 1: MMA update fused into the
        evaluation kernel, one launch per iteration; 0 (or False): separate update launch."""
        mode = 2 if mode is True else int(mode)
        self._chk(self._L.gtop_set_optimizer_fusion(self._h, mode))

    def set_optimizer_precision(self, dtype="f64"):
        """"f64" (default) or "f32": the arithmetic of the evaluations inside the batched optimizer (state, update
        and results stay fp64)."""
        self._chk(self._L.gtop_set_optimizer_precision(self._h, {"f64": GTOP_F64, "f32": GTOP_F32}[dtype]))

    def set_launch_geometry(self, waves=0, samples_per_lane=0):
        self._chk(self._L.gtop_set_launch_geometry(self._h, int(waves), int(samples_per_lane)))

    # -- evaluation --
    def eval_batch(self, x):
        """Host arrays in, host arrays out (fp64, PCIe copies included)."""
        x = _f64(x)
        B = x.shape[0]
        n = 9 * (self.m - 1)
        assert x.shape == (B, n)
        cost = np.empty(B)
        grad = np.empty((B, n))
        self._chk(self._L.gtop_eval_batch(self._h, B, _p(x), _p(cost), _p(grad)))
        return cost, grad

    def cost_nlopt(self, x, want_grad=True):
        """The nlopt_func-shaped entry point on trajectory 0."""
        x = _f64(x)
        g = np.empty_like(x) if want_grad else None
        c = self._L.gtop_cost_nlopt(x.size, _p(x), _p(g) if want_grad else None, self._h)
        if not np.isfinite(c):
            raise GtopError(1, self._L.gtop_last_error(self._h).decode())
        return c, g

    def eval_device(self, x, Df, T, cost=None, grad=None, stream=None):
        """torch CUDA tensors in HBM; launches on `stream` (default: torch's
        current stream) and returns without synchronising."""
        import torch
        dtype = {torch.float64: GTOP_F64, torch.float32: GTOP_F32}[x.dtype]
        B, n = x.shape
        m = n // 9 + 1
        assert n == 9 * (m - 1) and Df.numel() == B * 18
        stride = m if T.dim() == 2 else 0
        assert T.numel() == (B * m if stride else m)
        for t in (x, Df, T):
            assert t.is_cuda and t.is_contiguous() and t.dtype == x.dtype
        if cost is None:
            cost = torch.empty(B, dtype=x.dtype, device=x.device)
        if grad is None:
            grad = torch.empty(B, n, dtype=x.dtype, device=x.device)
        assert cost.is_contiguous() and grad.is_contiguous() and cost.numel() == B and grad.numel() == B * n
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        self._chk(self._L.gtop_eval_device(self._h, dtype, B, m, C.c_void_p(x.data_ptr()),
                                           C.c_void_p(Df.data_ptr()), C.c_void_p(T.data_ptr()), stride,
                                           C.c_void_p(cost.data_ptr()), C.c_void_p(grad.data_ptr()),
                                           C.c_void_p(stream)))
        return cost, grad

    def set_field_precisions(self, keep_fp32=True):
        """False: keep fp64 corner records only (the capturable map updates then skip the fp32 pass; fp32 evaluations fail)."""
        self._chk(self._L.gtop_set_field_precisions(self._h, 1 if keep_fp32 else 0))

    def push_rows(self, src, dst_ptrs, nbytes=None, stream=None, clock_minmax=None):
        """ONE kernel copies `src` (a contiguous CUDA tensor, or its first nbytes) to every device address in dst_ptrs
        (ints: slots in this process's and in peers' mapped buffers) — gtop_push_rows, the all-gather as stores."""
        import torch
        assert src.is_cuda and src.is_contiguous()
        if nbytes is None:
            nbytes = src.numel() * src.element_size()
        if stream is None:
            stream = torch.cuda.current_stream(src.device).cuda_stream
        arr = (C.c_void_p * len(dst_ptrs))(*[int(p) for p in dst_ptrs])
        self._chk(self._L.gtop_push_rows(self._h, C.c_void_p(src.data_ptr()), int(nbytes), arr, len(dst_ptrs),
                                         C.c_void_p(clock_minmax.data_ptr()) if clock_minmax is not None else None,
                                         C.c_void_p(stream)))

    def shared_alloc(self, nbytes):
        """(device address, 64-byte handle) of a zeroed allocation another process can map (gtop_shared_alloc)."""
        ptr = C.c_void_p()
        h = C.create_string_buffer(64)
        self._chk(self._L.gtop_shared_alloc(self._h, int(nbytes), C.byref(ptr), h))
        return int(ptr.value), bytes(h.raw)

    def shared_open(self, handle, owner_device=-1):
        """Device address, for this context's device, of a buffer a peer process made with shared_alloc (owner_device:
        that process's device ordinal, so that peer access is checked and enabled first)."""
        ptr = C.c_void_p()
        self._chk(self._L.gtop_shared_open(self._h, C.create_string_buffer(bytes(handle), 64), int(owner_device),
                                           C.byref(ptr)))
        return int(ptr.value)

    def shared_close(self, ptr):
        self._chk(self._L.gtop_shared_close(self._h, C.c_void_p(int(ptr))))

    def shared_free(self, ptr):
        self._chk(self._L.gtop_shared_free(self._h, C.c_void_p(int(ptr))))

    def clock_stamp(self, minmax, stream=None):
        """Enqueue a device-clock stamp: minmax (torch int64 tensor of 2 on the device, preset to [2**63 - 1, 0])
        becomes [min(., t), max(., t)] of the device's wall clock (gtop_device_clock_stamp)."""
        import torch
        assert minmax.is_cuda and minmax.dtype == torch.int64 and minmax.numel() == 2 and minmax.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(minmax.device).cuda_stream
        self._chk(self._L.gtop_device_clock_stamp(self._h, C.c_void_p(minmax.data_ptr()), C.c_void_p(stream)))

    def clock_hz(self):
        hz = C.c_double()
        self._chk(self._L.gtop_device_clock_hz(self._h, C.byref(hz)))
        return hz.value

    # -- setup / post-processing --
    TRAJ_STATS = ("time_sum", "length", "jerk", "mean_v", "max_v", "mean_a", "max_a", "acc_cost", "n_samples")

    def set_paths(self, waypoints, mean_v=1.8, init_time=0.3):
        """setPath for a batch: (B, m+1, 3) waypoints -> the context's problem; returns x0 (B, n)."""
        wp = _f64(waypoints)
        B, npts, _ = wp.shape
        m = npts - 1
        x0 = np.empty((B, 9 * (m - 1)))
        self._chk(self._L.gtop_set_paths(self._h, B, m, _p(wp), float(mean_v), float(init_time), _p(x0)))
        self.B, self.m = B, m
        return x0

    def get_problem(self):
        T = np.empty((self.B, self.m))
        Df = np.empty((self.B, 3, 6))
        self._chk(self._L.gtop_get_problem(self._h, _p(T), _p(Df)))
        return T, Df

    def trajectory_stats(self, x, dt_sample=0.01):
        """(coefficients (B, m, 18), stats (B, 9)) for the context's problem at free variables x."""
        x = _f64(x)
        B = x.shape[0]
        coeff = np.empty((B, self.m, 18))
        stats = np.empty((B, len(self.TRAJ_STATS)))
        self._chk(self._L.gtop_trajectory_stats(self._h, B, _p(x), float(dt_sample), _p(coeff), _p(stats)))
        return coeff, stats

    def trajectory_samples(self, x, dt_sample=0.01, max_samples=4096):
        """PolynomialTraj::getTraj for the context's problem at free variables x:
        (stats (B, 9), samples (B, max_samples, 3)); trajectory b has stats[b, 8] points."""
        x = _f64(x)
        B = x.shape[0]
        stats = np.empty((B, len(self.TRAJ_STATS)))
        samples = np.zeros((B, int(max_samples), 3))
        self._chk(self._L.gtop_trajectory_samples(self._h, B, _p(x), float(dt_sample), None, _p(stats), _p(samples),
                                                  int(max_samples)))
        return stats, samples

    # -- static field + moving boxes (EDTEnvironment) --
    def set_moving_boxes(self, p0, vel, scale):
        """Boxes {p0, vel, scale}, each (nbox, 3): centre p0 + vel*t, extent +-scale/2."""
        p0, vel, scale = (_f64(a).reshape(-1, 3) for a in (p0, vel, scale))
        assert p0.shape == vel.shape == scale.shape
        self._chk(self._L.gtop_set_moving_boxes(self._h, p0.shape[0], _p(p0), _p(vel), _p(scale)))

    def edt_query(self, pos, time):
        """(dist (N,), grad (N, 3)) at pos (N, 3), time (N,); time < 0 = static field only."""
        pos = _f64(pos).reshape(-1, 3)
        time = _f64(np.broadcast_to(time, (pos.shape[0],)))
        dist = np.empty(pos.shape[0])
        grad = np.empty((pos.shape[0], 3))
        self._chk(self._L.gtop_edt_query(self._h, pos.shape[0], _p(pos), _p(time), _p(dist), _p(grad)))
        return dist, grad

    def edt_coarse_query(self, pos, time):
        """EDTEnvironment::evaluateCoarseEDT: dist (N,) at pos (N, 3), time (N,); time < 0 = static field only."""
        pos = _f64(pos).reshape(-1, 3)
        time = _f64(np.broadcast_to(time, (pos.shape[0],)))
        dist = np.empty(pos.shape[0])
        self._chk(self._L.gtop_edt_coarse_query(self._h, pos.shape[0], _p(pos), _p(time), _p(dist)))
        return dist

    def edt_query_device(self, pos, time, stream=None):
        """torch fp64 CUDA tensors pos (N, 3), time (N,) -> dist (N,), grad (N, 3); asynchronous."""
        import torch
        assert pos.is_cuda and pos.dtype == torch.float64 and pos.is_contiguous() and time.is_contiguous()
        N = pos.shape[0]
        dist = torch.empty(N, dtype=torch.float64, device=pos.device)
        grad = torch.empty(N, 3, dtype=torch.float64, device=pos.device)
        if stream is None:
            stream = torch.cuda.current_stream(pos.device).cuda_stream
        self._chk(self._L.gtop_edt_query_device(self._h, N, C.c_void_p(pos.data_ptr()), C.c_void_p(time.data_ptr()),
                                                C.c_void_p(dist.data_ptr()), C.c_void_p(grad.data_ptr()),
                                                C.c_void_p(stream)))
        return dist, grad

    # -- batched optimizer --
    @staticmethod
    def default_bounds(waypoints, bos=3.0, vos=8.0, aos=10.0):
        """(B, m+1, 3) waypoints -> lb, ub of shape (B, n) (opti_node.launch bos/vos/aos)."""
        wp = _f64(waypoints)
        B, npts, _ = wp.shape
        m = npts - 1
        lb = np.empty((B, 9 * (m - 1)))
        ub = np.empty_like(lb)
        rc = load_library().gtop_default_bounds(B, m, _p(wp), bos, vos, aos, _p(lb), _p(ub))
        if rc != 0:
            raise GtopError(rc, "gtop_default_bounds")
        return lb, ub

    def optimize_batch(self, x0, lb, ub, max_evals):
        """Host arrays; returns (best x, its cost) for the problem of set_problem."""
        x = _f64(x0).copy()
        B = x.shape[0]
        lb, ub = _f64(lb), _f64(ub)
        assert lb.shape == x.shape == ub.shape
        cost = np.empty(B)
        self._chk(self._L.gtop_optimize_batch(self._h, B, _p(x), _p(lb), _p(ub), int(max_evals), _p(cost)))
        return x, cost

    def optimize_batch_ex(self, x0, lb, ub, max_evals, ftol_rel=0.0, xtol_rel=0.0, maxtime=0.0):
        """As optimize_batch, with NLopt's other stop rules; returns (x, cost, nevals, code)."""
        x = _f64(x0).copy()
        B = x.shape[0]
        lb, ub = _f64(lb), _f64(ub)
        cost = np.empty(B)
        nev = np.empty(B, dtype=np.int32)
        code = np.empty(B, dtype=np.int32)
        stop = GtopStop(int(max_evals), float(ftol_rel), float(xtol_rel), float(maxtime))
        ip = C.POINTER(C.c_int32)
        self._chk(self._L.gtop_optimize_batch_ex(self._h, B, _p(x), _p(lb), _p(ub), C.byref(stop), _p(cost),
                                                 nev.ctypes.data_as(ip), code.ctypes.data_as(ip)))
        return x, cost, nev, code

    def optimize_device_ex(self, x, Df, T, lb, ub, max_evals, ftol_rel=0.0, xtol_rel=0.0, maxtime=0.0, stream=None):
        """torch fp64 CUDA tensors; x is overwritten with the best point; returns (x, min_cost, nevals, code)."""
        import torch
        B, n = x.shape
        m = n // 9 + 1
        stride = m if T.dim() == 2 else 0
        for t in (x, Df, T, lb, ub):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
        min_cost = torch.empty(B, dtype=torch.float64, device=x.device)
        nev = torch.empty(B, dtype=torch.int32, device=x.device)
        code = torch.empty(B, dtype=torch.int32, device=x.device)
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        stop = GtopStop(int(max_evals), float(ftol_rel), float(xtol_rel), float(maxtime))
        self._chk(self._L.gtop_optimize_device_ex(
            self._h, B, m, C.c_void_p(x.data_ptr()), C.c_void_p(Df.data_ptr()), C.c_void_p(T.data_ptr()), stride,
            C.c_void_p(lb.data_ptr()), C.c_void_p(ub.data_ptr()), C.byref(stop), C.c_void_p(min_cost.data_ptr()),
            C.c_void_p(nev.data_ptr()), C.c_void_p(code.data_ptr()), C.c_void_p(stream)))
        return x, min_cost, nev, code

    def optimize_device(self, x, Df, T, lb, ub, max_evals, min_cost=None, stream=None):
        """torch fp64 CUDA tensors; x is overwritten with the best point."""
        import torch
        B, n = x.shape
        m = n // 9 + 1
        stride = m if T.dim() == 2 else 0
        for t in (x, Df, T, lb, ub):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float64
        if min_cost is None:
            min_cost = torch.empty(B, dtype=torch.float64, device=x.device)
        if stream is None:
            stream = torch.cuda.current_stream(x.device).cuda_stream
        self._chk(self._L.gtop_optimize_device(self._h, B, m, C.c_void_p(x.data_ptr()), C.c_void_p(Df.data_ptr()),
                                               C.c_void_p(T.data_ptr()), stride, C.c_void_p(lb.data_ptr()),
                                               C.c_void_p(ub.data_ptr()), int(max_evals),
                                               C.c_void_p(min_cost.data_ptr()), C.c_void_p(stream)))
        return x, min_cost

    # -- bookkeeping --
    def stats(self):
        it = C.c_int64()
        tt = C.c_double()
        self._chk(self._L.gtop_get_stats(self._h, C.byref(it), C.byref(tt)))
        return it.value, tt.value

    def reset_stats(self):
        self._chk(self._L.gtop_reset_stats(self._h))

    def cost_curve(self):
        cnt = C.c_int()
        self._chk(self._L.gtop_get_cost_curve(self._h, None, None, 0, C.byref(cnt)))
        c = np.empty(cnt.value)
        t = np.empty(cnt.value)
        if cnt.value:
            self._chk(self._L.gtop_get_cost_curve(self._h, _p(c), _p(t), cnt.value, C.byref(cnt)))
        return c, t

    def clear_cost_curve(self):
        self._chk(self._L.gtop_clear_cost_curve(self._h))


class Rendezvous:
    """N serial callers sharing one launch (include/gtop.h, gtop_rendezvous_*).  The context's problem must hold
    the N trajectories (row i = caller i).  Each caller thread uses `cost(i, x)` as its objective and calls
    `leave(i)` when its optimizer has returned."""

    def __init__(self, ctx, n_slots, m):
        self._L = load_library()
        self._ctx = ctx
        h = C.c_void_p()
        rc = self._L.gtop_rendezvous_create(C.byref(h), ctx._h, int(n_slots), int(m))
        if rc != 0:
            raise GtopError(rc, "gtop_rendezvous_create")
        self._h = h
        self.n = 9 * (m - 1)
        self._slots = [C.c_void_p(self._L.gtop_rendezvous_get_slot(h, i)) for i in range(n_slots)]

    def cost(self, i, x, want_grad=True):
        x = _f64(x)
        g = np.empty_like(x) if want_grad else None
        c = self._L.gtop_cost_nlopt_shared(x.size, _p(x), _p(g) if want_grad else None, self._slots[i])
        if not np.isfinite(c):
            raise GtopError(1, "gtop_cost_nlopt_shared: " + self._L.gtop_last_error(self._ctx._h).decode())
        return c, g

    def leave(self, i):
        return self._L.gtop_rendezvous_leave(self._slots[i])

    def set_timeout(self, seconds):
        """Longest a caller waits for the others; past it the rendezvous breaks for everybody (0 = for ever)."""
        self._L.gtop_rendezvous_set_timeout(self._h, float(seconds))

    def abort(self):
        """Wake every caller; all calls return an error from now on."""
        self._L.gtop_rendezvous_abort(self._h)

    def stats(self):
        n, cb, s = C.c_int64(), C.c_int64(), C.c_double()
        self._L.gtop_rendezvous_stats(self._h, C.byref(n), C.byref(s), C.byref(cb))
        return dict(launches=n.value, launch_seconds=s.value, callbacks=cb.value)

    def close(self):
        if getattr(self, "_h", None):
            self._L.gtop_rendezvous_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GtopGroup:
    """One batch over several devices from one process (include/gtop.h, gtop_group_*): the field replicated, the
    batch in contiguous slices, every slice launched on its own device, results gathered on the host or all-gathered
    on the devices (RCCL when the devices differ, peer copies otherwise)."""

    def __init__(self, devices, params=None):
        self._L = load_library()
        devs = (C.c_int * len(devices))(*[int(d) for d in devices])
        h = C.c_void_p()
        rc = self._L.gtop_group_create(C.byref(h), devs, len(devices))
        if rc != 0:
            raise GtopError(rc, self._L.gtop_group_last_error(None).decode())
        self._h = h
        self.devices = [int(d) for d in devices]
        self.set_params(**(params or {}))

    def _chk(self, rc):
        if rc != 0:
            raise GtopError(rc, self._L.gtop_group_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.gtop_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def gather_backend(self):
        return self._L.gtop_group_gather_backend(self._h).decode()

    def gather_note(self):
        """The backend and why: a fallback from RCCL to peer copies names its reason."""
        return self._L.gtop_group_gather_note(self._h).decode()

    def set_params(self, **kw):
        d = dict(OPTI_NODE_PARAMS)
        d.update(kw)
        p = GtopParams(**d)
        self._chk(self._L.gtop_group_set_params(self._h, C.byref(p)))

    def init_sdf_map(self, map_size, origin, resolution):
        self._chk(self._L.gtop_group_init_sdf_map(self._h, _p(_f64(map_size)), _p(_f64(origin)), float(resolution)))

    def update_sdf_map(self, pts):
        pts = _f64(pts).reshape(-1, 3)
        self._chk(self._L.gtop_group_update_sdf_map(self._h, _p(pts), pts.shape[0]))

    def update_sdf_map_window(self, min_pos, max_pos, pts):
        pts = _f64(pts).reshape(-1, 3)
        mn, mx = _f64(min_pos).reshape(3), _f64(max_pos).reshape(3)
        self._chk(self._L.gtop_group_update_sdf_map_window(self._h, _p(mn), _p(mx), _p(pts) if pts.size else None, pts.shape[0]))

    def set_problem(self, T, Df):
        Df = _f64(Df)
        B = Df.size // 18
        T = _f64(T)
        m, stride = (T.shape[1], T.shape[1]) if T.ndim == 2 else (T.shape[0], 0)
        self._chk(self._L.gtop_group_set_problem(self._h, B, m, _p(T), stride, _p(Df)))
        self.B, self.m = B, m

    def shards(self):
        out = []
        for i in range(len(self.devices)):
            a, b = C.c_int(), C.c_int()
            self._chk(self._L.gtop_group_shard(self._h, i, C.byref(a), C.byref(b)))
            out.append((a.value, b.value))
        return out

    def eval_batch(self, x):
        x = _f64(x)
        B, n = x.shape
        cost = np.empty(B)
        grad = np.empty((B, n))
        self._chk(self._L.gtop_group_eval_batch(self._h, B, _p(x), _p(cost), _p(grad)))
        return cost, grad

    def eval_resident(self, x=None, gather=1):
        """Evaluate the resident slices (x uploaded first when given) and all-gather on the devices; returns, per
        member, the gathered (cost, grad) as that device holds them (grad None unless gather == 2)."""
        if x is not None:
            x = _f64(x)
            self._chk(self._L.gtop_group_upload_x(self._h, x.shape[0], _p(x)))
        self._chk(self._L.gtop_group_eval_resident(self._h, int(gather), 1))
        n = 9 * (self.m - 1)
        out = []
        for i in range(len(self.devices)):
            c = np.empty(self.B)
            g = np.empty((self.B, n)) if gather == 2 else None
            self._chk(self._L.gtop_group_read_gathered(self._h, i, _p(c), _p(g) if g is not None else None))
            out.append((c, g))
        return out

    def set_launch_geometry(self, waves=0, samples_per_lane=0):
        """gtop_set_launch_geometry on every member's context (pins the kernel body: a slice and the whole batch then
        sum in the same order whatever their sizes)."""
        self._L.gtop_group_context.restype = C.c_void_p
        for i in range(len(self.devices)):
            ctx = C.c_void_p(self._L.gtop_group_context(self._h, i))
            rc = self._L.gtop_set_launch_geometry(ctx, int(waves), int(samples_per_lane))
            if rc != 0:
                raise GtopError(rc, self._L.gtop_last_error(ctx).decode())

    def launch_resident(self, x=None, gather=1):
        """eval_resident without waiting: the slices' evaluations and the all-gather are only ENQUEUED on the members'
        streams (gtop_group_eval_resident(synchronize = 0)); read the results with read_gathered() after synchronize()."""
        if x is not None:
            x = _f64(x)
            self._chk(self._L.gtop_group_upload_x(self._h, x.shape[0], _p(x)))
        self._chk(self._L.gtop_group_eval_resident(self._h, int(gather), 0))

    def synchronize(self):
        self._chk(self._L.gtop_group_synchronize(self._h))

    def read_gathered(self, member, grads=False):
        n = 9 * (self.m - 1)
        c = np.empty(self.B)
        g = np.empty((self.B, n)) if grads else None
        self._chk(self._L.gtop_group_read_gathered(self._h, int(member), _p(c), _p(g) if grads else None))
        return c, g

    def optimize_batch_ex(self, x0, lb, ub, max_evals, ftol_rel=0.0, xtol_rel=0.0, maxtime=0.0):
        x = _f64(x0).copy()
        B = x.shape[0]
        lb, ub = _f64(lb), _f64(ub)
        cost = np.empty(B)
        nev = np.empty(B, dtype=np.int32)
        code = np.empty(B, dtype=np.int32)
        stop = GtopStop(int(max_evals), float(ftol_rel), float(xtol_rel), float(maxtime))
        ip = C.POINTER(C.c_int32)
        self._chk(self._L.gtop_group_optimize_batch_ex(self._h, B, _p(x), _p(lb), _p(ub), C.byref(stop), _p(cost),
                                                       nev.ctypes.data_as(ip), code.ctypes.data_as(ip)))
        return x, cost, nev, code
