"""Generic executor, run-time half: compile the generated program once per layout variant, keep the
scratch buffers of the temporaries, fill the kernel-argument block and launch stage by stage through
the C ABI (``gt4mi_rtc_compile`` / ``gt4mi_module_*`` / ``gt4mi_launch`` in include/gt4py_amd.h).

Counterpart in the reference: the generated module's ``run()`` calling the JIT-built extension
(/root/reference/src/gt4py/cartesian/backend/gtc_common.py:144-168, :226-296).  Temporaries are owned by
the stencil object and cached per domain, where GridTools allocates them inside ``run_computation``.
"""

from __future__ import annotations

import ctypes
import weakref
from typing import Any, Dict, List, Tuple

import numpy as np

from . import hip_codegen
from ..stencil_object import StencilObject
from ... import _lib
from ...storage.allocators import PLACEMENT_MIN_BYTES, PLACEMENT_PERIOD, PLACEMENT_STEP

_U3 = ctypes.c_uint32 * 3


def _cache_dir():
    """Directory of compiled code objects, or None when caching is off.

    Counterpart of the reference's on-disk build cache (`.gt_cache`, cache_settings in
    /root/reference/src/gt4py/cartesian/config.py:45-63): there it holds generated sources and compiled
    extensions keyed by a stencil fingerprint; here one gfx950 code object per (generated source, options).
    ``GT4PY_AMD_CACHE_DIR`` moves it, ``GT4PY_AMD_CACHE_DIR=""`` (empty) switches it off."""
    import os
    import pathlib

    root = os.environ.get("GT4PY_AMD_CACHE_DIR")
    if root == "":
        return None
    path = pathlib.Path(root) if root else pathlib.Path.home() / ".cache" / "gt4py_amd" / "rtc"
    try:
        path.mkdir(parents=True, exist_ok=True)
    except OSError:
        return None
    return path


def _compile_cached(source: str, name: str, options) -> bytes:
    import hashlib
    import os

    cache = _cache_dir()
    entry = None
    if cache is not None:
        key = hashlib.sha256("\0".join([source, *options, "gfx950", str(_lib.GT4MI_ABI_VERSION)]).encode()).hexdigest()
        entry = cache / f"{key}.hsaco"
        try:
            code = entry.read_bytes()
            if code[:4] == b"\x7fELF":
                return code
        except OSError:
            pass
    code = _lib.rtc_compile(source, name, options)
    if entry is not None:
        try:
            tmp = entry.with_suffix(f".{os.getpid()}.tmp")
            tmp.write_bytes(code)
            os.replace(tmp, entry)  # atomic: concurrent processes never see a partial file
        except OSError:
            pass
    return code


class _Variant:
    """One compiled flavour of a generated program: loaded module + kernel handles."""

    def __init__(self, program: hip_codegen.GeneratedProgram, unit_i: bool, no_alias: bool):
        options = []
        if unit_i:
            options.append("-DGT4MI_UNIT_I_STRIDE=1")
        if no_alias:
            options.append("-DGT4MI_NO_ALIAS=1")
        lib = _lib.load()
        self.code = _compile_cached(program.source, f"{program.plan.stencil.name}.hip", options)
        self._code_buf = ctypes.create_string_buffer(self.code, len(self.code))
        module = ctypes.c_void_p()
        _lib.check("gt4mi_module_load", lib.gt4mi_module_load(self._code_buf, ctypes.byref(module)))
        self.module = module
        self.functions: List[ctypes.c_void_p] = []
        self.vec_functions: List[Any] = []  # the 16-byte-lane twin of a stage kernel, or None
        self.shared_functions: List[Any] = []  # ... and its form with temporaries shared between lanes (`_vecs`), or None
        self.tc_functions: List[Any] = []  # per kernel: the top-of-column-cache twins of a two-sweep column kernel
        for kern in program.kernels:
            fn = ctypes.c_void_p()
            _lib.check("gt4mi_module_function",
                       lib.gt4mi_module_function(module, kern.name.encode(), ctypes.byref(fn)))
            self.functions.append(fn)
            vfn = None
            if kern.vec and unit_i:
                vfn = ctypes.c_void_p()
                _lib.check("gt4mi_module_function",
                           lib.gt4mi_module_function(module, (kern.name + "_vec").encode(), ctypes.byref(vfn)))
            self.vec_functions.append(vfn)
            sfn = None
            if unit_i and no_alias and getattr(kern, "shared_halo", 0):
                sfn = ctypes.c_void_p()
                _lib.check("gt4mi_module_function",
                           lib.gt4mi_module_function(module, (kern.name + "_vecs").encode(), ctypes.byref(sfn)))
            self.shared_functions.append(sfn)
            tfns = []  # [(function, smallest domain K)], deepest cache first
            if kern.top_cache is not None and no_alias:  # emitted under GT4MI_NO_ALIAS only
                for n_reg, _, min_k in kern.top_cache:
                    tfn = ctypes.c_void_p()
                    _lib.check("gt4mi_module_function",
                               lib.gt4mi_module_function(module, f"{kern.name}_tc{n_reg}".encode(), ctypes.byref(tfn)))
                    scratch = ctypes.c_int(0)
                    _lib.check("gt4mi_function_info", lib.gt4mi_function_info(tfn, None, ctypes.byref(scratch), None))
                    if scratch.value == 0:  # else the register levels did not fit: spills cost more than the cache saves
                        tfns.append((tfn, int(min_k)))
            self.tc_functions.append(tfns)


#: scratch buffers (and the launch plans that point into them) are kept for this many HIP streams per stencil
MAX_SCRATCH_STREAMS = 4


def _ranges_disjoint(ranges: List[Tuple[int, int]]) -> bool:
    ranges = sorted(ranges)
    return all(ranges[n][1] <= ranges[n + 1][0] for n in range(len(ranges) - 1))


def _access_profile(plan):
    """(names the stencil writes, names it accesses anywhere at a non-zero or run-time offset); cached on the plan."""
    cached = getattr(plan, "_gt_access_profile", None)
    if cached is None:
        from .. import ir

        written, shifted = set(), set()
        for _, _, stmt in plan.stencil.statements():
            written.add(stmt.target.name)
            for e in (stmt.target, *ir.stmt_reads(stmt)):
                if isinstance(e, ir.FieldAccess) and (any(e.offset) or e.koffset is not None or e.absolute_k):
                    shifted.add(e.name)
        cached = (frozenset(written), frozenset(shifted))
        try:
            plan._gt_access_profile = cached
        except AttributeError:  # a frozen plan object: recompute next time
            pass
    return cached


def elements_disjoint(ptr_a: int, shape_a, ptr_b: int, shape_b, strides, itemsize: int) -> bool:
    """True when two arrays with the SAME byte strides provably share no element although their byte ranges overlap:
    interleaved slices of one buffer (``vel[..., 0]`` / ``vel[..., 1]``), the J halves of an I-contiguous parent.
    An element of ``a`` at index x is an element of ``b`` at index y iff ``sum (x - y) * stride == ptr_b - ptr_a``; with
    nested strides (each larger than everything the smaller ones can add up to) the differences are solved axis by axis
    from the largest stride down.  False = could not be shown (the caller treats the arrays as overlapping).  The same
    rule guards the kernel library's entry points (csrc/common.hip.h: elements_disjoint)."""
    if len(shape_a) != len(shape_b) or len(shape_a) != len(strides):
        return False
    delta = ptr_b - ptr_a
    axes = []  # (|stride|, lowest n, highest n) with n = x - y (sign folded into the bounds)
    for na, nb, st in zip(shape_a, shape_b, strides):
        if na <= 0 or nb <= 0:
            return True  # an empty array has no elements
        lo, hi = -(nb - 1), na - 1
        if st < 0:
            st, lo, hi = -st, -hi, -lo
        if st == 0:
            if lo or hi:
                return False
            continue
        axes.append((st, lo, hi))
    axes.sort()
    below, acc = [], 0
    for r, (st, lo, hi) in enumerate(axes):
        if r and st <= acc:
            return False  # strides not nested
        below.append(acc)
        acc += max(hi, -lo) * st
    if delta % itemsize or any(st % itemsize for st, _, _ in axes):
        return False  # elements could straddle each other

    def hit(r: int, rest: int) -> bool:
        if r < 0:
            return rest == 0
        st, lo, hi = axes[r]
        q = rest // st
        for n in (q, q + 1):
            if lo <= n <= hi and abs(rest - n * st) <= below[r] and hit(r - 1, rest - n * st):
                return True
        return False

    return not hit(len(axes) - 1, delta)


def _check_aliases(plan, names: List[str], spans: List[Tuple[int, int]], views, boxes=None) -> bool:
    """Arguments whose memory overlaps.  The reference's numpy backend evaluates a right-hand side completely before it
    assigns (npir_codegen.py:205-210), so a call like ``stencil(a, a)`` is well defined there.  The kernels here read
    and write concurrently and keep values in registers by NAME, so only what cannot depend on the order of evaluation
    is run: read-only arguments may overlap freely (returns True: nothing written is involved); a WRITTEN field may
    share its elements one to one with a read-only field when the stencil touches both at zero offsets only (returns
    False: the variant without ``__restrict__`` and without loads hoisted ahead of stores).  Everything else raises
    instead of returning a different answer."""
    written, shifted = _access_profile(plan)
    involved = False
    for a in range(len(names)):
        for b in range(a + 1, len(names)):
            na, nb = names[a], names[b]
            if not (spans[a][0] < spans[b][1] and spans[b][0] < spans[a][1]):
                continue
            if na not in written and nb not in written:
                continue
            involved = True
            w, x = (na, nb) if na in written else (nb, na)
            if views[na] != views[nb]:
                if boxes is not None and boxes[na][2:] == boxes[nb][2:] and elements_disjoint(
                        boxes[na][0], boxes[na][1], boxes[nb][0], boxes[nb][1], boxes[na][2], boxes[na][3]):
                    continue  # element-disjoint views of one buffer: the variant without __restrict__ runs them
                raise ValueError(f"fields '{na}' and '{nb}' overlap in memory without being the same elements, and the "
                                 f"stencil writes '{w}': the result would depend on the order of evaluation")
            if x in written or w in shifted or x in shifted:
                raise ValueError(f"fields '{na}' and '{nb}' are the same array, and the stencil "
                                 + (f"writes both" if x in written else f"writes '{w}' and reads at an offset")
                                 + ": the reference evaluates each right-hand side before it assigns, which kernels that "
                                   "read and write concurrently cannot reproduce -- pass separate arrays")
    return not involved


class HipGenericStencilObject(StencilObject):
    """StencilObject whose ``run`` launches run-time compiled gfx950 kernels, one per stage."""

    _gt_program_: hip_codegen.GeneratedProgram
    _gt_device_sync_: bool
    _gt_variants_: Dict[Tuple[bool, bool], _Variant]
    _gt_scratch_: Dict[Tuple[int, int, int], Any]
    _gt_launch_cache_: Dict[Any, Any]

    def _run_implementation(self, domain, origin, exec_info, arguments: Dict[str, Any]) -> None:
        cls = type(self)
        program = cls._gt_program_
        plan = program.plan
        lib = _lib.load()
        try:
            import torch

            stream = torch.cuda.current_stream().cuda_stream
        except Exception as ex:  # pragma: no cover - no GPU
            raise RuntimeError("hip:mi300 needs PyTorch-ROCm with a visible MI355X") from ex
        dI, dJ, dK = (int(d) for d in domain)
        # Launch plans are cached per (arrays, origins, domain, scalars): filling the argument block costs
        # 20-50 us of Python, a cached call ~10 us.  Arrays are remembered by identity through weak
        # references, so a recycled id() can never alias a dead array.
        try:
            # ... and per stream: temporaries live in a scratch buffer that concurrent streams must not share
            ckey = (stream, dI, dJ, dK, tuple(id(arguments[d.name]) for d in plan.api_fields),
                    tuple(tuple(origin[d.name]) for d in plan.api_fields), tuple(arguments[p.name] for p in plan.params))
            entry = cls._gt_launch_cache_.get(ckey)
        except TypeError:  # an unhashable scalar
            ckey = entry = None
        if entry is not None and all(r() is arguments[n] for n, r in entry[0]):
            _, args, launches, _keep = entry
        else:
            args, launches, _keep = self._prepare(arguments, origin, (dI, dJ, dK), stream)
            keep = _keep
            if ckey is not None:
                try:
                    refs = [(d.name, weakref.ref(arguments[d.name])) for d in plan.api_fields]
                    if len(cls._gt_launch_cache_) >= 8:
                        cls._gt_launch_cache_.pop(next(iter(cls._gt_launch_cache_)))
                    cls._gt_launch_cache_[ckey] = (refs, args, launches, keep)
                except TypeError:  # an array type without weak-reference support: do not cache
                    pass
        info = _lib.ExecInfo() if exec_info is not None else None
        info_ref = ctypes.byref(info) if info is not None else None
        t0 = t1 = None
        args_size = ctypes.sizeof(args)
        if len(launches) == 1:
            fn, grid, block, args_ref = launches[0]
            rc = lib.gt4mi_launch(fn, grid, block, args_ref, args_size, stream, info_ref)
            if rc:
                _lib.check("gt4mi_launch", rc)
        elif launches:  # every stage (and every K level of a plane-by-plane block) in one crossing of the C ABI
            n, fns, grids, blocks, arg_ptrs = _keep[2]
            rc = lib.gt4mi_launch_batch(n, fns, grids, blocks, arg_ptrs, args_size, stream, info_ref)
            if rc:
                _lib.check("gt4mi_launch_batch", rc)
        h0 = h1 = None
        if info is not None and launches:
            t0, t1 = info.run_cpp_start_time, info.run_cpp_end_time
            if info.run_hip_end_time > 0.0:
                h0, h1 = info.run_hip_start_time, info.run_hip_end_time
        if cls._gt_device_sync_:
            _lib.check("gt4mi_stream_sync", lib.gt4mi_stream_sync(stream))
        if exec_info is not None and t0 is not None:
            exec_info["run_cpp_start_time"] = t0
            exec_info["run_cpp_end_time"] = t1
            if h0 is not None:  # device-side interval of all stages of this call (hipEvent pair on the launch stream)
                exec_info["run_hip_start_time"] = h0
                exec_info["run_hip_end_time"] = h1

    def _prepare(self, arguments: Dict[str, Any], origin, domain, stream=None):
        """Argument block + launch list [(function, grid, block)] for one (arrays, origins, domain, scalars, stream)."""
        cls = type(self)
        program = cls._gt_program_
        plan = program.plan
        import torch

        dI, dJ, dK = domain
        args = program.args_struct()
        unit_i = True
        spans: List[Tuple[int, int]] = []
        views: Dict[str, Tuple[int, Tuple[int, ...], int]] = {}  # name -> (origin pointer, byte strides, itemsize)
        geometry: Dict[str, Tuple[int, int, int, int]] = {}  # name -> (origin pointer, sj, sk, itemsize)
        boxes: Dict[str, Tuple[int, Tuple[int, ...], Tuple[int, ...], int]] = {}  # name -> (base pointer, shape, byte strides, itemsize)
        for decl in plan.api_fields:
            arr = arguments[decl.name]
            c = hip_codegen._c_ident(decl.name)
            isz = arr.itemsize
            byte_strides = dict(zip(decl.axes, arr.strides))
            org = dict(zip(decl.axes, origin[decl.name]))
            if any(s % isz for s in arr.strides):
                raise ValueError(f"field '{decl.name}': strides {arr.strides} are not multiples of the item size")
            ptr = arr.ptr + sum(org[a] * byte_strides[a] for a in decl.axes)
            setattr(args, c, ptr)
            for axis, suffix in (("I", "si"), ("J", "sj"), ("K", "sk")):
                setattr(args, f"{c}_{suffix}", byte_strides.get(axis, 0) // isz)
            if "I" in decl.axes and byte_strides["I"] != isz:
                unit_i = False
            geometry[decl.name] = (ptr, byte_strides.get("J", 0) // isz, byte_strides.get("K", 0) // isz, isz)
            for dn, stride in enumerate(arr.strides[len(decl.axes):]):  # data dimensions
                setattr(args, f"{c}_d{dn}", stride // isz)
            hi = sum((n - 1) * s for n, s in zip(arr.shape, arr.strides) if s > 0) + isz
            lo = sum((n - 1) * s for n, s in zip(arr.shape, arr.strides) if s < 0)
            spans.append((arr.ptr + lo, arr.ptr + hi))
            views[decl.name] = (ptr, tuple(arr.strides), isz)
            boxes[decl.name] = (arr.ptr, tuple(arr.shape), tuple(arr.strides), isz)
        if plan.scratch:
            # One scratch buffer per (stream, domain): two calls of the stencil enqueued on different HIP streams run
            # concurrently and must not write each other's temporaries.  The buffer is allocated while `stream` is
            # the current stream, which is also the stream PyTorch's caching allocator orders its reuse on.
            key = (stream, dI, dJ, dK)
            entry = cls._gt_scratch_.get(key)
            if entry is None:
                layout, total = {}, 0
                temp_dims = {t.name: tuple(t.data_dims) for t in plan.stencil.temporaries}
                temp_levels = {t.name: (max(dK, 1) if "K" in t.axes else 1) for t in plan.stencil.temporaries}
                for name, (dt, ((ilo, ihi), (jlo, jhi))) in plan.scratch.items():
                    oi = -(-(-ilo) // 4) * 4  # the domain's first column on a 16-byte boundary
                    ni = -(-(dI + ihi + oi) // 32) * 32  # rows padded like the storage preset
                    nj = dJ + jhi - jlo
                    # equally shaped temporaries must not sit at the same address modulo 4 MiB (see
                    # storage/allocators.py:_placement_shift): temporary n starts at n * 1 MiB (mod 4 MiB) of a
                    # buffer whose base is brought to a 4 MiB boundary below
                    n_elem = int(np.prod(temp_dims.get(name, ()) or (1,)))  # data dimensions: outermost
                    nbytes = -(-(ni * nj * temp_levels[name] * n_elem * dt.itemsize) // 256) * 256
                    if nbytes >= PLACEMENT_MIN_BYTES:
                        total = -(-total // PLACEMENT_PERIOD) * PLACEMENT_PERIOD + (len(layout) % 4) * PLACEMENT_STEP
                    layout[name] = (total, ni, nj, dt.itemsize, oi, -jlo)
                    total += nbytes
                # zeros, not empty: a temporary that is assigned under a condition only is read (and ignored) where the
                # condition does not hold, and a byte that is neither 0 nor 1 read as a C++ bool is undefined behaviour
                # the compiler builds on -- stale memory made a mask temporary of an `elif` chain produce wrong stores
                # (found when such temporaries moved from registers to scratch).  Kernels only ever store 0 / 1 there.
                buf = torch.zeros(total + PLACEMENT_PERIOD, dtype=torch.uint8, device="cuda")
                # one domain at a time per stream: scratch can be gigabytes, and cached launch plans keep theirs alive
                for old_key in [k for k in cls._gt_scratch_ if k[0] == stream]:
                    del cls._gt_scratch_[old_key]
                for old_key in [k for k in cls._gt_launch_cache_ if k[0] == stream]:
                    del cls._gt_launch_cache_[old_key]
                # ... and a bounded number of streams (least recently created first): buffers made for short-lived
                # streams would otherwise never be released, and a recycled stream handle must not find a buffer the
                # caching allocator ordered on the stream that had the handle before
                streams = list(dict.fromkeys(k[0] for k in cls._gt_scratch_))
                for old_stream in streams[:max(0, len(streams) - (MAX_SCRATCH_STREAMS - 1))]:
                    for cache in (cls._gt_scratch_, cls._gt_launch_cache_):
                        for old_key in [k for k in cache if k[0] == old_stream]:
                            del cache[old_key]
                entry = cls._gt_scratch_[key] = (buf, layout)
            buf, layout = entry
            base = -(-buf.data_ptr() // PLACEMENT_PERIOD) * PLACEMENT_PERIOD
            for name, (off, ni, nj, isz, oi, oj) in layout.items():
                c = hip_codegen._c_ident(name)
                setattr(args, c, base + off + (oi + oj * ni) * isz)
                setattr(args, f"{c}_si", 1)
                setattr(args, f"{c}_sj", ni)
                decl = next(t for t in plan.stencil.temporaries if t.name == name)
                levels = max(dK, 1) if "K" in decl.axes else 1  # 2-d temporaries: one level, K stride 0
                setattr(args, f"{c}_sk", ni * nj if "K" in decl.axes else 0)
                dims = tuple(decl.data_dims)
                stride = ni * nj * levels
                for dn in range(len(dims) - 1, -1, -1):  # last data dimension varies fastest among them
                    setattr(args, f"{c}_d{dn}", stride)
                    stride *= dims[dn]
                geometry[name] = (base + off + (oi + oj * ni) * isz, ni, ni * nj if "K" in decl.axes else 0, isz)
        for p in plan.params:
            setattr(args, f"p_{hip_codegen._c_ident(p.name)}", np.dtype(p.dtype).type(arguments[p.name]).item())
        args.dI, args.dJ, args.dK = dI, dJ, dK
        args.k_lo, args.k_hi = 0, dK
        # How many items the origins of the API fields lie past a 16-byte boundary -- the same for all of them (the usual
        # case: storages allocated alike, one origin), else 0: the `_vec` strip kernels then start their lanes that far
        # before the domain (hip_codegen._emit_vector_kernel) instead of leaving such calls to the one-point-per-thread twin
        leads = {(geometry[d.name][0] % 16) // geometry[d.name][3] if geometry[d.name][3] in (4, 8) else -1
                 for d in plan.api_fields if "I" in d.axes}
        lead = leads.pop() if len(leads) == 1 else 0
        args.lead = lead if lead > 0 else 0

        no_alias = _ranges_disjoint(spans) or _check_aliases(plan, [d.name for d in plan.api_fields], spans, views, boxes)
        vkey = (unit_i, no_alias)
        variant = cls._gt_variants_.get(vkey)
        if variant is None:
            variant = cls._gt_variants_[vkey] = _Variant(program, *vkey)

        launches = []
        args_ref = ctypes.byref(args)
        keep_args = [args]
        per_level: Dict[int, Any] = {}  # K level -> copy of the argument block restricted to that level

        def level_args(k: int):
            if k not in per_level:
                copy = type(args).from_buffer_copy(args)
                copy.k_lo, copy.k_hi = k, k + 1
                keep_args.append(copy)
                per_level[k] = ctypes.byref(copy)
            return per_level[k]

        def geometry_of(kern, fn, vfn, tfns, sfn, levels: int):
            (ilo, ihi), (jlo, jhi) = kern.extent
            for tfn, min_k in tfns:  # deepest first: the most levels that stay in registers + LDS between the two sweeps
                if dK >= min_k:
                    fn = tfn
                    break
            ni, nj = dI + ihi - ilo, dJ + jhi - jlo
            if ni <= 0 or nj <= 0 or levels <= 0:
                return None
            nk = -(-levels // kern.k_per_thread) if kern.mapping == "ijk" else 1
            lanes = rows = 1
            if sfn is not None and no_alias and (kern.shared_preferred or vfn is None) and args.lead < kern.shared_vec and all(
                    (geometry[n][0] - args.lead * geometry[n][3]) % (kern.shared_vec * geometry[n][3]) == 0
                    and geometry[n][1] % kern.shared_vec == 0 and geometry[n][2] % kern.shared_vec == 0 for n in kern.shared_fields):
                # temporaries shared between lanes: waves overlap by the halo lanes
                per_wave = (64 - 2 * kern.shared_halo) * kern.shared_vec
                grid = _U3(-(-(ni + args.lead) // (per_wave * (kern.block[0] // 64))), -(-nj // (kern.block[1] * kern.shared_rows)), nk)
                return sfn, grid, _U3(*kern.block)
            if vfn is not None and no_alias and args.lead < kern.vec and all(
                    (geometry[n][0] - args.lead * geometry[n][3]) % (kern.vec * geometry[n][3]) == 0 and geometry[n][1] % kern.vec == 0
                    and geometry[n][2] % kern.vec == 0 for n in kern.vec_fields):
                fn, lanes, rows = vfn, kern.vec, kern.vec_rows  # every lane's vector is naturally aligned (after the lead)
                block = kern.vec_block or kern.block
                need = -(-(ni + args.lead) // lanes)  # lanes along I
                if block[1] == 1 and block[0] == 256 and -(-need // 320) * 320 < -(-need // 256) * 256:
                    block = (320, 1, 1)  # five waves: fewer idle lanes than a second workgroup per row
                return fn, _U3(-(-need // block[0]), -(-nj // (block[1] * rows)), nk), _U3(*block)
            grid = _U3(-(-ni // (kern.block[0] * lanes)), -(-nj // (kern.block[1] * kern.j_per_thread * rows)), nk)
            return fn, grid, _U3(*kern.block)

        triples = list(zip(program.kernels, variant.functions, variant.vec_functions, variant.tc_functions,
                           variant.shared_functions))
        n = 0
        while n < len(triples):
            kern = triples[n][0]
            if kern.plane is None:
                g = geometry_of(*triples[n], dK)
                if g is not None:
                    launches.append((*g, args_ref))
                n += 1
                continue
            # the stages of one sequential block with cross-column dependencies: once per K level, in sweep order
            m = n
            while m < len(triples) and triples[m][0].plane is not None and triples[m][0].plane[0] == kern.plane[0]:
                m += 1
            _, order, interval = kern.plane
            k0, k1 = interval.range(dK)
            levels = range(k0, k1) if order != "backward" else range(k1 - 1, k0 - 1, -1)
            body = [geometry_of(*t, 1) for t in triples[n:m]]
            for k in levels:
                for g in body:
                    if g is not None:
                        launches.append((*g, level_args(k)))
            n = m
        # the scratch buffer must outlive every cached plan that points into it
        n = len(launches)
        batch = None
        if n > 1:
            batch = (n, (ctypes.c_void_p * n)(*[f.value if isinstance(f, ctypes.c_void_p) else f for f, _, _, _ in launches]),
                     (ctypes.c_uint32 * (3 * n))(*[v for _, g, _, _ in launches for v in g]),
                     (ctypes.c_uint32 * (3 * n))(*[v for _, _, b, _ in launches for v in b]),
                     (ctypes.c_void_p * n)(*[ctypes.addressof(a._obj) for _, _, _, a in launches]))
        return args, launches, (cls._gt_scratch_.get((stream, dI, dJ, dK)), keep_args, batch)
