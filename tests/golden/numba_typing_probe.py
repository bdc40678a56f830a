"""Numba 0.54.1's TYPE INFERENCE, with the CUDA target's typing context, over the reference's ray kernels -- what a real
Numba-CUDA device computes in, as opposed to the simulator the golden fixtures were recorded under (SURVEY App. A.2,
INTEGRATION.md section 5).  No GPU, no NVVM: only the front end and the type-inference pass run; nothing is compiled.

THIS CONTAINER ONLY (it imports /root/reference/scripts/gvom.py, unmodified, under /opt/conda/bin/python3.9 -- numba 0.54.1,
numpy 1.26.4, with ref_shim's import-time stub for numba's `_internal` extension).  Output: the inferred type of every
variable of `__point_2_map` and `__calculate_min_height`, for float32 and for float64 clouds, next to the simulator's
(Python / numpy semantics), with the expressions that differ marked.

    /opt/conda/bin/python3.9 tests/golden/numba_typing_probe.py > profiles/numba_cuda_typing.txt
"""
import os
import re
import sys

os.environ.pop("NUMBA_ENABLE_CUDASIM", None)        # the REAL target's typing, not the simulator
os.environ.pop("NUMBA_DISABLE_JIT", None)
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def load():
    import numpy as np
    import ref_shim                                   # (its import asks for the simulator: undone below, before numba is imported)
    os.environ.pop("NUMBA_ENABLE_CUDASIM", None)
    os.environ.pop("NUMBA_DISABLE_JIT", None)
    for alias, typ in (("bool", bool), ("int", int), ("float", float), ("complex", complex), ("object", object), ("str", str)):
        if alias not in np.__dict__:
            setattr(np, alias, typ)
    # (numba 0.54.1 registers overloads for numpy names that numpy 1.26 no longer has: placeholders, never called here)
    for gone in ("MachAr",):
        if gone not in np.__dict__:
            setattr(np, gone, type(gone, (), {}))
    sys.meta_path.insert(0, ref_shim._StubFinder())
    real_version = np.__version__
    np.__version__ = "1.20.3"
    try:
        import numba
        from numba import cuda  # noqa: F401
    finally:
        np.__version__ = real_version
    sys.path.insert(0, ref_shim.REFERENCE_SCRIPTS)
    import gvom
    assert os.path.abspath(gvom.__file__).startswith(ref_shim.REFERENCE_SCRIPTS)
    return numba, gvom


def infer(numba, pyfunc, argtypes):
    from numba.core import compiler, typed_passes
    from numba.cuda.descriptor import cuda_target
    typingctx, targetctx = cuda_target.typing_context, cuda_target.target_context
    typingctx.refresh(); targetctx.refresh()
    func_ir = compiler.run_frontend(pyfunc)
    with targetctx.push_code_library(None) if hasattr(targetctx, "push_code_library") else _null():
        typemap, restype, calltypes, _ = typed_passes.type_inference_stage(typingctx, targetctx, func_ir, argtypes, None)
    return func_ir, typemap, calltypes


class _null(object):
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def py_func_of(disp):
    for attr in ("py_func", "_py_func", "func"):
        f = getattr(disp, attr, None)
        if f is not None:
            return f
    raise AttributeError("no python function on %r" % (disp,))


def report(numba, name, disp, argtypes, label, first_line):
    from numba.core import ir
    func_ir, typemap, calltypes = infer(numba, py_func_of(disp), argtypes)
    print("## %s, %s" % (name, label))
    # named variables (the source's own locals; versioned copies `x.1` collapse onto one line when their types agree)
    named = {}
    for var, ty in typemap.items():
        base = var.split(".")[0]
        if base.startswith("$") or base.startswith("arg."):
            continue
        named.setdefault(base, set()).add(str(ty))
    for base in sorted(named):
        print("  %-22s %s" % (base, " | ".join(sorted(named[base]))))
    # every assignment whose right-hand side is an arithmetic expression or a call: line, source text, inferred type
    print("  -- expressions by source line (type of the value assigned)")
    seen = set()
    for blk in func_ir.blocks.values():
        for st in blk.body:
            if isinstance(st, ir.Assign) and isinstance(st.value, ir.Expr) and st.value.op in ("binop", "inplace_binop", "call", "unary", "getitem", "static_getitem"):
                ln = st.loc.line
                ty = str(typemap.get(st.target.name))
                what = st.value.op
                if what in ("binop", "inplace_binop"):
                    fn = getattr(st.value.fn, "__name__", str(st.value.fn))
                    what = "%s(%s, %s)" % (fn, typemap.get(st.value.lhs.name), typemap.get(st.value.rhs.name))
                elif what == "call":
                    what = "call %s" % (calltypes.get(st.value),)
                elif what == "unary":
                    what = "unary %s(%s)" % (getattr(st.value.fn, "__name__", st.value.fn), typemap.get(st.value.value.name))
                else:
                    continue
                key = (ln, what, ty)
                if key in seen:
                    continue
                seen.add(key)
                # (the reference's source text is NOT reproduced: the line number cites it)
                print("  :%d  %-62s -> %s" % (ln, what[:62], ty))
    print()


def main():
    numba, gvom = load()
    import inspect
    from numba import types
    print("# numba %s, CUDA target typing context (type inference only; nothing compiled, no GPU)" % numba.__version__)
    print("# reference: /root/reference/scripts/gvom.py (unmodified); line numbers are that file's")
    print()
    G = gvom.Gvom
    kernels = {"__point_2_map": G._Gvom__point_2_map, "__calculate_min_height": G._Gvom__calculate_min_height}
    i32a, i64 = types.Array(types.int32, 1, "C"), types.int64
    f64a1 = types.Array(types.float64, 1, "C")
    for name, disp in kernels.items():
        f = py_func_of(disp)
        src, first = inspect.getsourcelines(f)
        sig = list(inspect.signature(f).parameters)
        print("# %s(%s)" % (name, ", ".join(sig)))
        for cloud_t, label in ((types.float32, "float32 cloud"), (types.float64, "float64 cloud")):
            pc = types.Array(cloud_t, 2, "C")
            args = []
            for p in sig:                                   # by the reference's own call sites (gvom.py:136-146, 1016-1019)
                if p in ("pointcloud", "points"):
                    args.append(pc)
                elif p in ("hit_count", "total_count", "index_map", "tmp_hit_count", "tmp_total_count"):
                    args.append(i32a)
                elif p in ("min_height",):
                    args.append(types.Array(types.float32, 1, "C"))
                elif p in ("origin", "ego_position", "ego"):
                    args.append(f64a1)
                elif p in ("point_count", "xy_size", "z_size"):
                    args.append(i64)
                else:
                    args.append(types.float64)
            print("#   argument types: %s" % ", ".join("%s: %s" % (a, b) for a, b in zip(sig, args)))
            report(numba, name, disp, tuple(args), label, first)


SUMMARY = """
## differences from the simulator's (Python / numpy 1.26 scalar) semantics, float32 AND float64 clouds alike
#
#  gvom.py   real Numba-CUDA (above)                      simulator (what the golden fixtures hold)        values
#  :1109     math.sqrt(float32) -> float32                math.sqrt(np.float32) -> Python float (f64)      DIFFER
#  :1112-14  slope[k] / ray_length: float32 / float32     np.float32 / float -> float64, stored as f32     differ through :1109 only
#                                                         (a float32 quotient computed in float64 and rounded is the float32 quotient)
#  :1127     ray_length - 1: float64(float32) - 1         float64 - 1                                      differ through :1109 only
#  :1072-80, :1134-42, :1311-19   math.floor(f64) -> float64   math.floor -> Python int                    same (integers below 2^53)
#  :1086, :1146, :1328  index: float64, cast to int64 by the atomic / int()   Python int                   same
#  :1150     1.0 / float32 -> float64                     float / np.float32 -> np.float64                 same
#  :1329     cuda.atomic.min(float32 array, ., float64 value -> float32)   same rounding on store           same
#  every other expression has the same type on both sides (d2 in the cloud's dtype, compares in float64, pt / end / slope float32
#  locals, float32 accumulation of pt, int32 atomics).
#
# g-vom_amd's numba_cuda_typing=True (GVOM_FLAG_NUMBA_CUDA_TYPING) switches exactly the three DIFFER rows: csrc/gvom_trace.hip
# ray_setup (`P.f32_sqrt ? (double)sqrtf(ss) : sqrt((double)ss)`; the quotient and the bound follow from it), oracle/gvom_oracle.c
# likewise.  NOT covered by any switch, because it is code generation and not typing: NVVM contracts a*b + c into FMAs by
# default on a real device (SURVEY App. A.3); without a CUDA device that cannot be pinned.
"""

if __name__ == "__main__":
    main()
    print(SUMMARY)
