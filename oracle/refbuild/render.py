#!/usr/bin/env python
"""Renders the reference's OWN CPU rulebook code into a compilable C++ file.

TEST INFRASTRUCTURE (like everything under oracle/): nothing in the product path imports this.

The reference's C++ does not exist as files: `spconv/csrc/sparse/indices.py` holds it as string
fragments that the `pccm` code generator assembles at build time, against headers of the `cumm`
package (PyPI `cumm`, pinned `>=0.7.11,<0.8.0` in the reference's setup.py:42-44).  Neither pccm nor
cumm is installed here and the reference's build system is not run.  What this script does instead:

  1. registers small stand-in modules for `pccm` and the handful of `cumm` names indices.py imports
     (just enough to record arguments and code fragments -- no code generation logic of pccm is
     reproduced, the fragments are written out in the order the reference emits them);
  2. loads `/root/reference/spconv/csrc/sparse/indices.py` FROM WHERE IT LIES and calls the
     reference's own generator classes `ConvOutLocIter` (indices.py:76-269) and
     `SparseConvIndicesCPU` (indices.py:1620-1778) for ndim = 1..4;
  3. writes the collected C++ to `oracle/_ref/ref_indices.cpp` (git-ignored; never committed).

The emitted function bodies -- the hash loops, the output-coordinate algebra, the first-seen
numbering, the SubM mirror trick -- are the reference's text, verbatim.  What is NOT the reference's:
the `tv::` value types and cumm's `ConvProblem` / `TensorGeneric` layout, restated in
`tv_shim.h` from their documented behaviour (plain row-major index arithmetic), and the extern "C"
entry points appended at the end.  `oracle/Makefile` target `ref` compiles the result into
`oracle/_ref/libspconv_ref.so`.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = os.environ.get("SPCONV_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_ref")


# ----------------------------------------------------------------- stand-in `pccm`
class FunctionCode:
    def __init__(self):
        self.args, self.targs, self.blocks, self.inits = [], [], [], []
        self.ret_type = "void"

    def arg(self, names, typ, default=None, **_):
        for n in names.split(","):
            self.args.append((n.strip(), typ, default))
        return self

    def targ(self, name):
        self.targs.append(f"typename {name}")
        return self

    def nontype_targ(self, name, typ):
        self.targs.append(f"{typ} {name}")
        return self

    def raw(self, text):
        self.blocks.append(text)
        return self

    def ctor_init(self, name, value):
        self.inits.append((name, value))
        return self

    def ret(self, typ, *_, **__):
        self.ret_type = typ
        return self


def _mark(kind):
    def deco(*dargs, **dkw):
        def wrap(fn, meta=dkw):
            if kind == "pybind" and getattr(fn, "_pccm", None) is not None:
                return fn                      # a binding annotation on top of static / member: keep the inner kind
            fn._pccm = dict(kind=kind, **meta)
            return fn
        if len(dargs) == 1 and callable(dargs[0]) and not dkw:      # bare @decorator
            return wrap(dargs[0], {})
        return wrap
    return deco


class Class:
    def __init__(self, *a, **k):
        self._members, self._includes, self._param_classes = [], [], []

    @property
    def class_name(self):
        return type(self).__name__

    def add_dependency(self, *a):
        pass

    def add_include(self, *names):
        self._includes += list(names)

    def add_param_class(self, ns, obj, alias=None):
        self._param_classes.append((ns, obj, alias))

    def add_member(self, name, typ, *a, **k):
        self._members.append((name, typ))

    def add_static_const(self, *a, **k):
        pass

    def functions(self):
        """(python name, meta, FunctionCode) of every decorated generator method, source order."""
        out = []
        for klass in reversed(type(self).__mro__):
            for name, fn in vars(klass).items():
                meta = getattr(fn, "_pccm", None)
                if meta is not None:
                    out.append((name, meta, fn))
        return out


pccm = types.ModuleType("pccm")
pccm.FunctionCode = FunctionCode
pccm.Class = pccm.ParameterizedClass = Class
pccm.member_function = _mark("member")
pccm.static_function = _mark("static")
pccm.constructor = _mark("ctor")
pccm.destructor = _mark("dtor")
pccm.external_function = _mark("external")
pccm.literal = lambda v: str(v).lower() if isinstance(v, bool) else str(v)
pccm.cuda = types.ModuleType("pccm.cuda")
pccm.cuda.cuda_global_function = _mark("cuda_global")
pccm.cuda.static_function = _mark("cuda_static")
pccm.cuda.member_function = _mark("cuda_member")
pccm.pybind = types.ModuleType("pccm.pybind")
pccm.pybind.mark = _mark("pybind")
pccm.pybind.mark_prop_getter = _mark("pybind")
pccm.pybind.mark_prop_setter = _mark("pybind")
pccm.pybind.PybindClassMixin = type("PybindClassMixin", (), {"add_pybind_member": lambda self, *a, **k: None})
pccm.boolean = lambda v: "true" if v else "false"


# ----------------------------------------------------------------- stand-in `cumm` names
class DType:
    def __init__(self, name):
        self.name = name

    def __str__(self):
        return self.name

    def __format__(self, spec):
        return self.name


dtypes = types.ModuleType("cumm.dtypes")
dtypes.DType = DType
dtypes.int32 = DType("int32_t")
dtypes.int64 = DType("int64_t")
dtypes.float32 = DType("float")


class TensorGeneric(Class):
    """cumm.gemm.layout.TensorGeneric: a row-major ndim layout; restated in tv_shim.h."""

    def __init__(self, ndim, fast_divmod=False, dtype=None):
        super().__init__()
        self.ndim, self.index_t = ndim, str(dtype or dtypes.int32)

    def cxx(self):
        return f"tvshim::TensorGeneric<{self.ndim}, {self.index_t}>"


class ConvProblem(Class):
    """cumm.conv.params.ConvProblem: the geometry record; restated in tv_shim.h."""

    def __init__(self, ndim, *a, **k):
        super().__init__()
        self.ndim = ndim

    def cxx(self):
        return f"tvshim::ConvProblem<{self.ndim}>"


def _dispatch_ints(code, ints, var):
    for v in ints:
        code.raw(f"if ({var} == {v}) {{")
        yield v
        code.raw("}")


def _install():
    mods = {"pccm": pccm, "pccm.cuda": pccm.cuda, "pccm.pybind": pccm.pybind}
    for name in ("cumm", "cumm.gemm", "cumm.gemm.core", "cumm.gemm.core.metaarray", "cumm.gemm.layout",
                 "cumm.gemm.codeops", "cumm.common", "cumm.conv", "cumm.conv.params", "cumm.constants"):
        mods[name] = types.ModuleType(name)
    mods["cumm.dtypes"] = dtypes
    mods["cumm"].dtypes = dtypes
    mods["cumm.gemm.core.metaarray"].MetaArray = object
    mods["cumm.gemm.core.metaarray"].seq = lambda *a: list(a)
    mods["cumm.gemm.layout"].TensorGeneric = TensorGeneric
    mods["cumm.gemm.layout"].to_stride = lambda s: s
    for n in ("TensorView", "TensorViewHashKernel", "TensorViewKernel", "ThrustLib", "GemmDTypes", "GemmBasic"):
        setattr(mods["cumm.common"], n, type(n, (Class,), {}))
    # maxpool.py's imports (its GPU classes are defined at import and never called here)
    for name in ("cumm.gemm.mask_iters", "cumm.gemm.thread_map", "spconv.csrc.utils", "spconv.csrc.utils.launch"):
        mods[name] = types.ModuleType(name)
    mods["cumm.gemm.mask_iters"].MaskTileIterator = type("MaskTileIterator", (Class,), {})
    mods["cumm.gemm.mask_iters"].MaskTileIteratorParams = type("MaskTileIteratorParams", (Class,), {})
    mods["cumm.gemm"].thread_map = mods["cumm.gemm.thread_map"]
    mods["spconv.csrc.utils.launch"].LaunchUtils = type("LaunchUtils", (Class,), {})
    co = mods["cumm.gemm.codeops"]
    co.dispatch_ints = _dispatch_ints
    co.unpack = lambda name, rng, left="[", right="]": ", ".join(f"{name}{left}{i}{right}" for i in rng)
    co.unpack_str = lambda name, rng, sep="_": ", ".join(f"{name}{sep}{i}" for i in rng)
    mods["cumm.gemm"].codeops = co
    mods["cumm.conv.params"].ConvProblem = ConvProblem
    mods["cumm.constants"].CUMM_CPU_ONLY_BUILD = True
    # gather.py imports OMPLib from the reference's own package: a stand-in module keeps the package's
    # __init__ (which loads the compiled extension) from being imported
    for name in ("spconv", "spconv.csrc", "spconv.csrc.sparse", "spconv.csrc.sparse.cpu_core"):
        mods[name] = types.ModuleType(name)
    mods["spconv.csrc.sparse.cpu_core"].OMPLib = type("OMPLib", (Class,), {})
    for name in ("spconv", "spconv.csrc", "spconv.csrc.sparse", "spconv.csrc.utils"):
        mods[name].__path__ = []          # packages: `from ..utils.launch import LaunchUtils` (maxpool.py:29)
    sys.modules.update(mods)


def load_reference_indices():
    _install()
    path = os.path.join(REF, "spconv", "csrc", "sparse", "indices.py")
    spec = importlib.util.spec_from_file_location("_reference_indices", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, path


def load_reference_gather():
    """spconv/csrc/sparse/gather.py (GatherCPU: the row gather / scatter-add of ConvAlgo.Native's CPU path)."""
    _install()
    path = os.path.join(REF, "spconv", "csrc", "sparse", "gather.py")
    spec = importlib.util.spec_from_file_location("_reference_gather", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod, path


def load_reference_module(fname: str, modname: str):
    """spconv/csrc/sparse/<fname> loaded FROM WHERE IT LIES under the package name it has in the reference
    (its relative imports then resolve against the stand-in packages)."""
    _install()
    path = os.path.join(REF, "spconv", "csrc", "sparse", fname)
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod, path


def emit_static_functions(obj, namespace: str, names):
    """The named static functions of a generator class, bodies verbatim, inside `namespace`."""
    out = [f"namespace {namespace} {{"]
    seen = []
    for pyname, meta, fn in obj.functions():
        if meta["kind"] != "static" or pyname not in names:
            continue
        code = fn(obj)
        out.append(f"{code.ret_type} {_signature(pyname, code)} {{")
        out += code.blocks
        out.append("}")
        seen.append(pyname)
    assert sorted(seen) == sorted(names), (seen, names)
    out.append(f"}}  // namespace {namespace}")
    return "\n".join(out)


def emit_gather_class(obj):
    out = ["namespace refgather {"]
    for pyname, meta, fn in obj.functions():
        if meta["kind"] != "static":
            continue
        code = fn(obj)
        out.append(f"{code.ret_type} {_signature(pyname, code)} {{")
        out += code.blocks
        out.append("}")
    out.append("}  // namespace refgather")
    return "\n".join(out)


# ----------------------------------------------------------------- emission
def _signature(name, code: FunctionCode, with_defaults=True):
    parts = []
    for n, t, d in code.args:
        parts.append(f"{t} {n}" + (f" = {d}" if (d is not None and with_defaults) else ""))
    return f"{name}({', '.join(parts)})"


def emit_lociter(obj, struct_name):
    """One ConvOutLocIter instance -> a struct with the reference's members and method bodies."""
    lines = [f"struct {struct_name} {{"]
    aliases = {alias: o for _, o, alias in obj._param_classes if alias}
    lines.append(f"  using ConvProblem = {aliases['ConvProblem'].cxx()};")
    lines.append(f"  using LayoutNPQ = {aliases['LayoutNPQ'].cxx()};")
    lines.append(f"  using LayoutRS = {aliases['LayoutRS'].cxx()};")
    for n, t in obj._members:
        lines.append(f"  {t} {n};")
    for pyname, meta, fn in obj.functions():
        code = fn(obj)
        tmpl = f"  template <{', '.join(code.targs)}>\n" if code.targs else ""
        if meta["kind"] == "ctor":
            init = ", ".join(f"{n}({v})" for n, v in code.inits)
            lines.append(f"{tmpl}  {_signature(struct_name, code)} : {init} {{")
        else:
            cname = meta.get("name") or pyname
            ret = code.ret_type.replace(obj.class_name, struct_name)
            const = " const" if meta.get("const") else ""
            lines.append(f"{tmpl}  {ret} {_signature(cname, code)}{const} {{")
        lines += code.blocks
        lines.append("  }")
    lines.append("};")
    return "\n".join(lines)


def emit_cpu_class(obj, ndim):
    ns = f"refnd{ndim}"
    out = [f"namespace {ns} {{"]
    out.append(emit_lociter(obj.loc_iter, "ConvLocIter"))
    out.append(emit_lociter(obj.loc_iter_64, "ConvLocIter64"))
    out.append(f"using ConvProblem = tvshim::ConvProblem<{ndim}>;")
    for pyname, meta, fn in obj.functions():
        if meta["kind"] != "static":
            continue
        code = fn(obj)
        out.append(f"{code.ret_type} {_signature(pyname, code)} {{")
        out += code.blocks
        out.append("}")
    out.append(f"}}  // namespace {ns}")
    return "\n".join(out)


C_API = r'''
// ---- extern "C" entry points (NOT reference text): plain pointers in, reference functions called ----
namespace {
template <int ND> tv::array<int, ND> arr(const int *p) {
  tv::array<int, ND> a;
  for (int i = 0; i < ND; ++i) a[i] = p[i];
  return a;
}
}  // namespace
#define REF_DISPATCH(ND)                                                                              \
  case ND: {                                                                                           \
    tv::Tensor inds = tv::from_blob(indices, {n_in, ND + 1});                                          \
    tv::Tensor pairs = tv::from_blob(indice_pairs, {2, kv, pair_cap});                                 \
    tv::Tensor outi = tv::from_blob(out_inds, {out_cap, ND + 1});                                      \
    tv::Tensor num = tv::from_blob(indice_num_per_loc, {kv});                                          \
    if (subm)                                                                                          \
      return refnd##ND::generate_subm_conv_inds(inds, pairs, outi, num, batch_size, arr<ND>(in_dims),  \
                                                arr<ND>(ksize), arr<ND>(dilation));                    \
    return refnd##ND::generate_conv_inds(inds, pairs, outi, num, batch_size, arr<ND>(out_dims),        \
                                         arr<ND>(in_dims), arr<ND>(ksize), arr<ND>(stride),            \
                                         arr<ND>(padding), arr<ND>(dilation), transposed != 0);        \
  }
extern "C" int ref_generate_inds(int ndim, int subm, int transposed, int32_t *indices, int n_in,
                                 int32_t *indice_pairs, int pair_cap, int32_t *out_inds, int out_cap,
                                 int32_t *indice_num_per_loc, int batch_size, const int *in_dims,
                                 const int *out_dims, const int *ksize, const int *stride,
                                 const int *padding, const int *dilation) {
  int kv = 1;
  for (int i = 0; i < ndim; ++i) kv *= ksize[i];
  try {
    switch (ndim) {
      REF_DISPATCH(1) REF_DISPATCH(2) REF_DISPATCH(3) REF_DISPATCH(4)
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "reference code raised: %s\n", e.what());
    return -2;
  }
  return -1;
}

// GatherCPU::gather / scatter_add (gather.py:30-86) on fp32 rows
extern "C" int ref_gather(float *out, float *in, int32_t *inds, int nhot, int n_in, int channel) {
  tv::Tensor o = tv::from_blob(out, {nhot, channel}), i = tv::from_blob(in, {n_in, channel});
  tv::Tensor x = tv::from_blob(inds, {nhot});
  refgather::gather(o, i, x);
  return 0;
}
extern "C" int ref_scatter_add(float *out, float *in, int32_t *inds, int nhot, int n_out, int channel) {
  tv::Tensor o = tv::from_blob(out, {n_out, channel}), i = tv::from_blob(in, {nhot, channel});
  tv::Tensor x = tv::from_blob(inds, {nhot});
  refgather::scatter_add(o, i, x);
  return 0;
}
'''


C_API_8F = r'''
// ---- extern "C" entry points of the section-8f code (NOT reference text) ----
// Point2VoxelCPU::point_to_voxel[_empty_mean]_static on fp32 points, 3-d, zyx index order.  Buffers as the class
// constructor prepares them (pointops.py:568-578): voxels / indices / num_per_voxel zeroed [max_voxels, ...],
// densehashdata = the grid filled with -1.  Returns the number of voxels.
extern "C" int ref_point2voxel(float *points, int n, int nfeat, float *voxels, int32_t *indices,
                               int32_t *num_per_voxel, int32_t *densehash, int64_t *points_voxel_id,
                               const float *vsize, const int *grid_size, const int *grid_stride,
                               const float *coors_range, int max_voxels, int max_points, int empty_mean) {
  tv::Tensor pts = tv::from_blob(points, {n, nfeat});
  tv::Tensor vox = tv::from_blob(voxels, {max_voxels, max_points, nfeat});
  tv::Tensor ind = tv::from_blob(indices, {max_voxels, 3});
  tv::Tensor num = tv::from_blob(num_per_voxel, {max_voxels});
  tv::Tensor grid = tv::from_blob(densehash, {grid_size[0], grid_size[1], grid_size[2]});
  tv::Tensor pid = tv::from_blob(points_voxel_id, {n}, 8);
  std::array<float, 3> vs{vsize[0], vsize[1], vsize[2]};
  std::array<int, 3> gs{grid_size[0], grid_size[1], grid_size[2]}, gst{grid_stride[0], grid_stride[1], grid_stride[2]};
  std::array<float, 6> cr{coors_range[0], coors_range[1], coors_range[2], coors_range[3], coors_range[4], coors_range[5]};
  try {
    auto res = empty_mean ? refp2v3::point_to_voxel_empty_mean_static(pts, vox, ind, num, grid, pid, vs, gs, gst, cr, true)
                          : refp2v3::point_to_voxel_static(pts, vox, ind, num, grid, pid, vs, gs, gst, cr, true);
    return static_cast<int>(std::get<0>(res).dim(0));
  } catch (const std::exception &e) {
    std::fprintf(stderr, "reference code raised: %s\n", e.what());
    return -2;
  }
}

// IndiceMaxPoolCPU::forward / backward (maxpool.py:620-700) on fp32 rows: ONE kernel offset's pair list
extern "C" int ref_maxpool_fwd(float *out, float *in, int32_t *out_inds, int32_t *in_inds, int nhot, int n_out,
                               int n_in, int channel) {
  refpool::forward(tv::from_blob(out, {n_out, channel}), tv::from_blob(in, {n_in, channel}),
                   tv::from_blob(out_inds, {nhot}), tv::from_blob(in_inds, {nhot}));
  return 0;
}
extern "C" int ref_maxpool_bwd(float *out, float *in, float *dout, float *din, int32_t *out_inds, int32_t *in_inds,
                               int nhot, int n_out, int n_in, int channel) {
  refpool::backward(tv::from_blob(out, {n_out, channel}), tv::from_blob(in, {n_in, channel}),
                    tv::from_blob(dout, {n_out, channel}), tv::from_blob(din, {n_in, channel}),
                    tv::from_blob(out_inds, {nhot}), tv::from_blob(in_inds, {nhot}));
  return 0;
}
extern "C" int ref_global_pool_rearrange(int32_t *out_indices, int32_t *coords, int32_t *counts, int nhot, int ncol,
                                         int batch) {
  refpool::global_pool_rearrange(tv::from_blob(out_indices, {batch, nhot}), tv::from_blob(coords, {nhot, ncol}),
                                 tv::from_blob(counts, {batch}));
  return 0;
}
'''


def main():
    mod, path = load_reference_indices()
    os.makedirs(OUT_DIR, exist_ok=True)
    parts = ["// GENERATED by oracle/refbuild/render.py from " + path,
             "// Function bodies are the reference's own text (Apache-2.0, Copyright 2021 Yan Yan).",
             "// Build product only: lives under oracle/_ref/ (git-ignored), never committed.",
             '#include "tv_shim.h"', ""]
    for ndim in (1, 2, 3, 4):
        problem = ConvProblem(ndim)
        cpu = mod.SparseConvIndicesCPU(problem, dtypes.int32)
        parts.append(emit_cpu_class(cpu, ndim))
    gmod, gpath = load_reference_gather()
    parts.append("// ---- from " + gpath)
    parts.append(emit_gather_class(gmod.GatherCPU()))
    # the CPU voxeliser (pointops.py:493-766) and the CPU max-pool loops
    # (maxpool.py:590-703), rendered the same way
    pmod, ppath = load_reference_module("pointops.py", "spconv.csrc.sparse.pointops")
    parts.append("// ---- from " + ppath)
    parts.append(emit_static_functions(pmod.Point2VoxelCPU(dtypes.float32, 3, True), "refp2v3",
                                       ["point_to_voxel_static", "point_to_voxel_empty_mean_static"]))
    mmod, mpath = load_reference_module("maxpool.py", "spconv.csrc.sparse.maxpool")
    parts.append("// ---- from " + mpath)
    parts.append(emit_static_functions(mmod.IndiceMaxPoolCPU(), "refpool",
                                       ["forward", "backward", "global_pool_rearrange"]))
    parts.append(C_API)
    parts.append(C_API_8F)
    out = os.path.join(OUT_DIR, "ref_indices.cpp")
    with open(out, "w") as f:
        f.write("\n".join(parts))
    print("wrote", out)


if __name__ == "__main__":
    main()
