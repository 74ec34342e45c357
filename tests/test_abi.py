"""The shipped C-ABI library loads on a CPU-only box and exports every symbol that
include/ndgpu_nextcorrect.h declares (no compute calls here)."""
import os
import re

import util


def _declared_functions(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"^[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", txt, flags=re.M):
        names.add(m.group(1))
    return names


def test_exports(native_lib):
    header = os.path.join(os.path.dirname(util.HERE), "include", "ndgpu_nextcorrect.h")
    names = _declared_functions(header)
    assert {"nextCorrect", "free_consensus_trimed", "align", "align_hq", "ndgpu_correct_batch",
            "ndgpu_correct_piles", "ndgpu_db_create", "poa_to_consensus", "malloc_vd", "revcomp_bseq"} <= names
    for n in sorted(names):
        assert hasattr(native_lib, n), "missing export: " + n


def test_host_helpers(native_lib):
    import ctypes as C
    b = C.create_string_buffer(b"ACGTNacgtMK")
    native_lib.revcomp_bseq(b, 11)
    assert b.value == b"MKacgtNACGT"
    native_lib.reverse_str(b, 11)
    assert b.value == b"TGCANtgcaKM"
    c = C.create_string_buffer(b"acgt")
    native_lib.str_toupper(c)
    assert c.value == b"ACGT"
    native_lib.str_tolower(c)
    assert c.value == b"acgt"


def test_device_count_does_not_need_gpu(native_lib):
    assert native_lib.ndgpu_device_count() >= 0
