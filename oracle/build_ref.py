"""Build the reference's own CPU extensions from the sources where they lie.

Test infrastructure only.  Nothing is copied into this repo; outputs go only under
oracle/_ref/ (git-ignored).  The GPU box has no /root/reference; nothing there calls this.

fpemu_cpp -- /root/reference/mpemu/pytquant/cpp/fpemu_impl.cpp, the reference's flags
(setup.py:12-17 of the reference: -mf16c -march=native -mlzcnt -fopenmp, plus -O3), twice:

  oracle/_ref/native/fpemu_ref_native.so   AVX-512 intrinsic routines for the size/32*32 prefix
  oracle/_ref/scalar/fpemu_ref_scalar.so   + -mno-avx512f: the scalar routines everywhere

SURVEY.md section 8(a) says which build is canonical for which mode.

cmodel -- simple_gemm_dev / simple_conv2d_dev (setup.py:33-45 of the reference), needed to
import the reference's Python package (mpemu/__init__.py imports cmodel) and as the oracle of
the "next" emulated-GEMM row.
"""
import os
import sys

REF_ROOT = "/root/reference"
REF_SRC = REF_ROOT + "/mpemu/pytquant/cpp/fpemu_impl.cpp"
REF_CMODEL = REF_ROOT + "/mpemu/cmodel/simple"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")

_BASE_FLAGS = ["-mf16c", "-march=native", "-mlzcnt", "-fopenmp", "-O3"]
_CMODEL_FLAGS = ["-march=native", "-fopenmp", "-O3"]


def have_reference():
    return os.path.isfile(REF_SRC)


def _load(name, sources, flags, include=None, verbose=False):
    from torch.utils import cpp_extension

    if not have_reference():
        raise RuntimeError("reference sources not present under " + REF_ROOT)
    bdir = os.path.join(OUT, name)
    os.makedirs(bdir, exist_ok=True)
    return cpp_extension.load(
        name=name,
        sources=sources,
        extra_cflags=flags,
        extra_ldflags=["-fopenmp"],
        extra_include_paths=include or [],
        build_directory=bdir,
        verbose=verbose,
    )


def load(variant, verbose=False):
    """JIT-build (or reuse) one fpemu_cpp variant; returns the module (forward/backward)."""
    flags = list(_BASE_FLAGS)
    if variant == "scalar":
        flags.append("-mno-avx512f")
    elif variant != "native":
        raise ValueError(variant)
    bdir = os.path.join(OUT, variant)
    os.makedirs(bdir, exist_ok=True)
    from torch.utils import cpp_extension

    if not have_reference():
        raise RuntimeError("reference sources not present under " + REF_ROOT)
    return cpp_extension.load(
        name="fpemu_ref_" + variant,
        sources=[REF_SRC],
        extra_cflags=flags,
        extra_ldflags=["-fopenmp"],
        build_directory=bdir,
        verbose=verbose,
    )


def load_cmodel(which, verbose=False, nofma=False):
    """which: 'simple_gemm_dev' | 'simple_conv2d_dev' (module names the reference imports).

    With the reference's own flags (-march=native; g++ defaults to -ffp-contract=fast) GCC contracts
    the engine's _mm256_add_ps(c, _mm256_mul_ps(a, b)) into vfmadd231ps on an FMA host: the accumulation
    is an fmaf chain.  nofma=True adds -mno-fma -mno-avx512f (AVX-512F carries its own EVEX fused
    multiply-add): the same sources as a host without FMA compiles them, i.e. separate vmulps / vaddps
    roundings.  Both are "the reference"; DESIGN.md says which kernel variant pins which."""
    srcs = {
        "simple_gemm_dev": ["simple_gemm.cpp", "simple_gemm_impl.cpp", "simple_mm_engine.cpp"],
        "simple_conv2d_dev": ["simple_conv2d.cpp", "simple_conv2d_impl.cpp", "simple_mm_engine.cpp"],
    }[which]
    name = which + ("_nofma" if nofma else "")
    flags = _CMODEL_FLAGS + (["-mno-fma", "-mno-avx512f"] if nofma else [])
    return _load(name, [os.path.join(REF_CMODEL, s) for s in srcs], flags, include=[REF_CMODEL], verbose=verbose)


def import_reference_package(variant="native"):
    """Import the reference's Python package `mpemu` from /root/reference (CPU only).

    Registers the JIT-built extensions under the module names the package imports
    (fpemu_cpp, simple_gemm_dev, simple_conv2d_dev).  Writes no bytecode into the read-only tree.
    """
    sys.dont_write_bytecode = True
    sys.modules["fpemu_cpp"] = load(variant)
    sys.modules["simple_gemm_dev"] = load_cmodel("simple_gemm_dev")
    sys.modules["simple_conv2d_dev"] = load_cmodel("simple_conv2d_dev")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import mpemu  # noqa: F401  (the reference package)

    if not mpemu.__file__.startswith(REF_ROOT):
        raise RuntimeError("a different `mpemu` is already imported: " + mpemu.__file__)
    return mpemu


def main():
    if not have_reference():
        print("oracle/_ref: reference tree absent, nothing built")
        return 0
    v = "-v" in sys.argv
    for variant in ("native", "scalar"):
        m = load(variant, verbose=v)
        print("oracle/_ref/%s: built %s" % (variant, m.__file__))
    for which in ("simple_gemm_dev", "simple_conv2d_dev"):
        m = load_cmodel(which, verbose=v)
        print("oracle/_ref/%s: built %s" % (which, m.__file__))
    for which in ("simple_gemm_dev", "simple_conv2d_dev"):
        m = load_cmodel(which, verbose=v, nofma=True)
        print("oracle/_ref/%s_nofma: built %s" % (which, m.__file__))
    return 0


if __name__ == "__main__":
    sys.exit(main())
