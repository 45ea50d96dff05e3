"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/litho_abbe.h declares, its host-only entry points agree with the golden vectors, and
bad arguments come back as error codes (no compute call is made without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest

from helpers import ROOT, WL

HEADER = os.path.join(ROOT, "include", "litho_abbe.h")


@pytest.fixture(scope="module")
def nat():
    from lithographysimulator_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", ROOT, "-j", "8", "all"])
    return _native


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(litho_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_entry_points():
    names = declared_symbols()
    for must in ("litho_abbe_accumulate", "litho_abbe_field", "litho_source_bitmap", "litho_source_compact",
                 "litho_pupil", "litho_pupil_stack", "litho_postprocess", "litho_mask_spectrum", "litho_abbe_workspace_bytes",
                 "litho_epsilon_n"):
        assert must in names


def test_library_exports_every_declared_symbol(nat):
    lib = ctypes.CDLL(nat.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/litho_abbe.h but not exported"
    assert sorted(nat.exported_symbols()) == declared_symbols()


def test_plan_record_layout_matches_the_header(nat):
    """litho_abbe_plan: 16 plan words + valid + pn + N + planes, all int32 (ctypes mirror: _native.PlanRecord)."""
    text = open(HEADER).read()
    assert "int32_t words[16];" in text and "int32_t valid;" in text and "int32_t pn, N, planes;" in text
    assert ctypes.sizeof(nat.PlanRecord) == 20 * 4
    assert [f[0] for f in nat.PlanRecord._fields_] == ["words", "valid", "pn", "N", "planes"]
    # a NULL record is an argument error, not a crash (no GPU call is made)
    assert nat.lib().litho_abbe_accumulate_planned(None, None, 1, None, None, 0, 256, 512, None, None, 0, None, None, None) == nat.E_ARG


def test_options_record_layout_matches_the_header(nat):
    """litho_abbe_options: `size` then int32 fields in the header's order (ctypes mirror: _native.Options); unset = -1."""
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    body = re.search(r"typedef struct litho_abbe_options \{(.*?)\} litho_abbe_options;", text, flags=re.S).group(1)
    fields = [n.strip() for decl in re.findall(r"int32_t\s+([^;]+);", body) for n in decl.split(",")]
    assert fields == [f[0] for f in nat.Options._fields_]
    assert ctypes.sizeof(nat.Options) == 4 * len(fields)
    o = nat.Options.make({"coarse": 2, "batch": 7})
    assert o.size == ctypes.sizeof(nat.Options) and o.coarse == 2 and o.batch == 7 and o.tile == -1 and o.poison == -1
    with pytest.raises(KeyError):
        nat.Options.make({"no_such_knob": 1})
    # argument errors before any GPU work: a NULL image, and an options record with an impossible size
    lib = nat.lib()
    assert lib.litho_abbe_accumulate_opts(None, None, 1, None, None, 0, 256, 512, None, None, 0, None, None, None, None) == nat.E_ARG


def test_engine_options_blocks_nest_and_leave_the_environment_alone(nat):
    before = dict(os.environ)
    assert nat.current_options() is None
    with nat.engineOptions(coarse=2, batch=5):
        with nat.engineOptions(batch=9):
            o = nat.current_options({"tile": 8})
            assert (o.coarse, o.batch, o.tile, o.groups) == (2, 9, 8, -1)
        o = nat.current_options()
        assert (o.coarse, o.batch, o.tile) == (2, 5, -1)
    assert nat.current_options() is None and dict(os.environ) == before
    with pytest.raises(KeyError):
        with nat.engineOptions(bogus=1):
            pass


def test_target_arch_and_version(nat):
    assert nat.lib().litho_target_arch() == b"gfx950"
    assert nat.lib().litho_version() >= 100


def test_epsilon_n_matches_reference_table(nat, golden):
    for pn, ps, eps, N in golden("g3_mask_spectra.npz")["sizing_table"]:
        e, n = nat.epsilon_n(4 / pn, ps, WL)
        assert n == int(N) and e == eps


def test_postprocess_size(nat):
    lib = nat.lib()
    for pn, expect in ((64, 64), (256, 256), (1024, 1024), (2048, 2048), (4096, 4094), (8192, 8192)):
        eps, _ = nat.epsilon_n(4 / pn, 25, WL)
        out = ctypes.c_int(0)
        assert lib.litho_postprocess_size(pn, eps, ctypes.byref(out)) == 0
        assert out.value == expect


def test_argument_errors_are_codes_not_crashes(nat):
    lib = nat.lib()
    nbytes = ctypes.c_size_t(0)
    assert lib.litho_abbe_workspace_bytes(2048, 4096, ctypes.byref(nbytes)) == 0 and nbytes.value > 2048 * 2048 * 8
    assert lib.litho_abbe_workspace_bytes(2047, 4096, ctypes.byref(nbytes)) == nat.E_ARG        # odd pn
    assert lib.litho_abbe_workspace_bytes(64, 100, ctypes.byref(nbytes)) == nat.E_ARG           # N not 2^k
    assert lib.litho_abbe_workspace_bytes(64, 32, ctypes.byref(nbytes)) == nat.E_NSMALL         # N < pn (Q6)
    assert lib.litho_abbe_workspace_bytes(64, 128, None) == nat.E_ARG
    assert lib.litho_abbe_accumulate(None, None, 1, None, 1, 64, 128, None, None, 0, None) == nat.E_ARG
    assert lib.litho_abbe_field(None, None, 64, 128, None, None, 0, None) == nat.E_ARG
    assert lib.litho_source_bitmap(7, 0.0, 0.5, 64, 0.0, 0.0, 4, 0.0, None, None) == nat.E_ARG
    assert lib.litho_abbe_last_plan(None) == nat.E_ARG
    with pytest.raises(RuntimeError):
        nat.check(nat.E_NSMALL, "x")
    with pytest.raises(IndexError):
        nat.check(nat.E_INDEX, "x")
    with pytest.raises(ValueError):
        nat.check(nat.E_ARG, "x")


def test_pupil_length_4_is_an_index_error_before_any_gpu_work(nat):
    # J == 4 is rejected on the host (pupil.py:91-92 indexes [4]); no device is touched
    coeffs = (ctypes.c_uint16 * 4)(0, 0, 0, 0x3C00)
    assert nat.lib().litho_pupil(coeffs, 4, 64, 0.7, 193.0, 0, None, ctypes.c_void_p(8), None) == nat.E_INDEX


def test_pupil_stack_argument_errors_before_any_gpu_work(nat):
    # litho_pupil_stack: `ab[4] = d` on a vector shorter than 5 is the reference loop's IndexError; nulls / empty stacks are E_ARG
    lib = nat.lib()
    c4, c5, d = (ctypes.c_uint16 * 4)(), (ctypes.c_uint16 * 5)(), (ctypes.c_uint16 * 2)(0x5640, 0xD640)
    out = ctypes.c_void_p(8)
    assert lib.litho_pupil_stack(c4, 4, d, 2, 64, 0.7, 193.0, None, out, None) == nat.E_INDEX
    assert lib.litho_pupil_stack(c5, 5, d, 0, 64, 0.7, 193.0, None, out, None) == nat.E_ARG
    assert lib.litho_pupil_stack(c5, 5, None, 2, 64, 0.7, 193.0, None, out, None) == nat.E_ARG
    assert lib.litho_pupil_stack(None, 5, d, 2, 64, 0.7, 193.0, None, out, None) == nat.E_ARG
    assert lib.litho_pupil_stack(c5, 5, d, 2, 64, 0.7, 193.0, None, None, None) == nat.E_ARG


def test_diag_switches_do_not_compile_into_the_product():
    """LITHO_DIAG_* switches remove loads / stores / barriers from the kernels (timing diagnostics, wrong results).
    A stray -D in a product build must fail to compile; only LITHO_DIAG_BUILD (scripts/build_variants.sh, separate
    output, library reports "gfx950-diag") may carry them."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(ROOT, "lithographysimulator_amd", "csrc", "inst_04.hip")
    base = [hipcc, "--offload-arch=gfx950", "-std=c++17", "-fsyntax-only", "--cuda-host-only", src]
    bad = subprocess.run(base + ["-DLITHO_DIAG_XNOSTORE"], capture_output=True, text=True)
    assert bad.returncode != 0 and "LITHO_DIAG_" in bad.stderr
    ok = subprocess.run(base + ["-DLITHO_DIAG_XNOSTORE", "-DLITHO_DIAG_BUILD"], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr[-1500:]
