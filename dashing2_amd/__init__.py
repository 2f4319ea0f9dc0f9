"""dashing2_amd -- MI355X-native implementation of dashing2's two data-parallel hot paths.

The product is ``libd2g.so`` (hand-written HIP kernels for gfx950 + an x86 host half) behind
the C ABI declared in ``include/d2g.h``, and the ``dashing2`` drop-in CLI built from
``dashing2_amd/host``.  This Python package is only a thin ctypes mirror of that C ABI, used
by the tests, ``bench.py`` and the multi-GPU launcher (``dashing2_amd.dist``); it never
re-implements any part of the path and it fails loudly when the shared object is missing.
"""
from .capi import (  # noqa: F401
    D2GError, Context, CmpSet, SeqPack, Comm, AllPairs, lib, build, LIB_PATH,
    comm_unique_id, allpairs_step_all, allpairs_prepare_all, bcast_sigs,
    SIMILARITY, CONTAINMENT, SYMMETRIC_CONTAINMENT, POISSON_LLR, INTERSECTION, UNION_SIZE,
    CMP_AUTO, CMP_DIRECT, CMP_BITSLICE, BITSLICE_OPS_PER_GROUP_EXTRA, TIME_K1, TIME_K2, TIME_K2PREP, TIME_K3, TIME_K0, PinnedArray,
    wang_hash, seed_mask, oph_xor_const, oph_m, oph_finalize, densify, epilogue_lut,
    epilogue_gtlt, epilogue_neq, host_epilogue_ut, operand_layout, ut_count, ut_partition,
)

__all__ = [
    "D2GError", "Context", "CmpSet", "SeqPack", "Comm", "AllPairs", "lib", "build", "LIB_PATH",
    "comm_unique_id", "allpairs_step_all", "allpairs_prepare_all", "bcast_sigs",
    "SIMILARITY", "CONTAINMENT", "SYMMETRIC_CONTAINMENT", "POISSON_LLR", "INTERSECTION", "UNION_SIZE",
    "CMP_AUTO", "CMP_DIRECT", "CMP_BITSLICE", "BITSLICE_OPS_PER_GROUP_EXTRA", "TIME_K1", "TIME_K2", "TIME_K2PREP", "TIME_K3", "TIME_K0", "PinnedArray",
    "wang_hash", "seed_mask", "oph_xor_const", "oph_m", "oph_finalize", "densify", "epilogue_lut",
    "epilogue_gtlt", "epilogue_neq", "host_epilogue_ut", "operand_layout", "ut_count", "ut_partition",
]
