"""paffy_amd -- MI355X (gfx950) implementation of paffy's per-record PAF/CIGAR hot path.

The product is the C-ABI library `libpaffy_hip.so` (include/paffy_hip.h); this package is the
Python host mirror used by the tests and bench.py. There is no CPU fallback: importing the
engine without the built library, or without a GPU, raises.
"""
from .engine import (ADD_MISMATCHES, FILTER, INVERT, PASS, REMOVE_MISMATCHES, SHATTER, STATS, TRIM_ENDS, TRIM_FIXED, TRIM_IDENTITY, Engine, PafError,
                     PlanInfo, Stage, add_mismatches, build_library, chain, dedupe, filter, invert, library_path, pipe, shatter, stage, stage_trim_ends, tile, trim)

__all__ = ["Engine", "Stage", "PlanInfo", "PafError", "stage", "stage_trim_ends", "pipe", "invert", "shatter", "trim", "add_mismatches", "tile", "chain", "filter", "dedupe", "build_library", "library_path",
           "INVERT", "TRIM_IDENTITY", "TRIM_FIXED", "SHATTER", "ADD_MISMATCHES", "REMOVE_MISMATCHES", "PASS", "FILTER", "TRIM_ENDS", "STATS"]
