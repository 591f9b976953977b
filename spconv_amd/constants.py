"""Flags of the path (subset of spconv/constants.py:30-121 that still has a meaning here)."""
import os

# spconv/constants.py:119-121: sort output rows by mask so that whole tiles skip offsets (reference
# default "1").  ops.get_indice_pairs_implicit_gemm keeps that meaning for its do_sort argument.
SPCONV_DO_SORT = os.getenv("SPCONV_DO_SORT", "1") == "1"
# The layer modules: unset (default) = the density-aware rows layout built inside the rulebook build
# (ops.rows_layout: classified and regrouped on the device, no sort, nothing read back -- the drop-in
# analogue of the reference's default); "1" = the reference's explicit mask sort of every rulebook
# (radix argsort + tables copied into tile order: ~100 us per rulebook); "0" = rows stay in input order.
MODULE_DO_SORT = {"1": True, "0": False}.get(os.getenv("SPCONV_DO_SORT", ""), "layout")
# Row order of a strided layer's outputs in the layer modules.  The reference fixes none on the GPU (sort + unique of
# the coordinate keys, all.py:1533-1552, or hash-slot order, indices.py:1380-1425); its CPU path is first-seen
# (indices.py:1742-1771).  "sorted" (default) = ascending coordinate key through the level's rank map (a cheaper build,
# SubM layers behind it without a hash table, x-neighbours in adjacent rows for every gather of the level);
# "first_seen" = the CPU reference's numbering.  The functional API (ops.get_indice_pairs*) is first-seen always.
CONV_OUTPUT_ORDER = os.getenv("SPCONV_AMD_CONV_ORDER", "sorted")
# spconv/constants.py:36: layout of checkpoints produced by spconv 1.x / 2.1 ("KRSC", "RSKC", "RSCK").
SAVED_WEIGHT_LAYOUT = os.getenv("SPCONV_SAVED_WEIGHT_LAYOUT", "")
# spconv/constants.py:112: skip the constructor checks of SparseConvTensor while torch.fx traces a
# model (its arguments are Proxies then)
SPCONV_FX_TRACE_MODE = os.getenv("SPCONV_FX_TRACE_MODE", "0") == "1"
# "Identical to the reference" where the reference's own behaviour is a slip: =1 reproduces (a) the average-pool
# backward that MULTIPLIES by the window count (csrc/sparse/maxpool.py:262-300; the default divides: the derivative
# of the forward) and (b) the CPU voxeliser's mean fill whose accumulator is carried from voxel to voxel
# (csrc/sparse/pointops.py:663-686: `mean_value.clear()`; the default fills with the voxel's own mean).  The
# native library reads the same variable for (a).
REFERENCE_QUIRKS = os.getenv("SPCONV_AMD_REFERENCE_QUIRKS", "0") == "1"
ALL_WEIGHT_IS_KRSC = True
FILTER_HWIO = False


class AllocKeys:
    """Names of the buffers the reference allocator hands out (spconv/constants.py:66-98)."""
    PairFwd = "PairFwd"
    PairBwd = "PairBwd"
    PairMask = "PairMask"
    PairMaskBwd = "PairMaskBwd"
    MaskArgSort = "MaskArgSort"
    MaskArgSortBwd = "MaskArgSortBwd"
    OutIndices = "OutIndices"
    IndiceNumPerLoc = "IndiceNumPerLoc"
    OutFeatures = "OutFeatures"
    DIn = "DIn"
    DFilters = "DFilters"
