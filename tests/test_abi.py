"""The shipped C-ABI library loads on a CPU-only box and exports every symbol that
include/ndgpu_nextcorrect.h declares (no compute calls here)."""
import os
import re

import util


def _declared_functions(header):
    txt = open(header).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    names = set()
    for m in re.finditer(r"^(?!typedef\b)[A-Za-z_][\w\s\*]*?\b(\w+)\s*\([^;{]*\)\s*;", txt, flags=re.M):   # (function-pointer typedefs are not functions)
        names.add(m.group(1))
    return names


def test_exports(native_lib):
    header = os.path.join(os.path.dirname(util.HERE), "include", "ndgpu_nextcorrect.h")
    names = _declared_functions(header)
    assert {"nextCorrect", "free_consensus_trimed", "align", "align_hq", "ndgpu_correct_batch",
            "ndgpu_correct_piles", "ndgpu_db_create", "poa_to_consensus", "malloc_vd", "revcomp_bseq"} <= names
    for n in sorted(names):
        assert hasattr(native_lib, n), "missing export: " + n


def _dynamic_exports(path):
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}


def test_nothing_but_the_abi_is_exported():
    """The libraries export the functions their headers declare and nothing else (runtime globals, template instantiations and
    kernel stubs stay local: a process may load them next to the reference's ovlseq.so / other libraries)."""
    from nextdenovo_amd import build
    root = os.path.dirname(util.HERE)
    for lib, header in ((build.LIB, "ndgpu_nextcorrect.h"), (build.OVL_LIB, "ndgpu_overlap.h")):
        exp = {n for n in _dynamic_exports(lib) if not n.startswith("__hip_cuid_")}
        assert exp == _declared_functions(os.path.join(root, "include", header)), (lib, sorted(exp ^ _declared_functions(os.path.join(root, "include", header))))


def test_align_nd_matches_reference(native_lib):
    """align_nd (lib/align.c:580-679, exported by the reference's nextcorrect.so, no caller): same strings on random pairs."""
    import ctypes as C
    import random
    import pytest
    import refpipe
    if not refpipe.have_ref("nextcorrect.so"):
        pytest.skip("compiled reference not built")
    ref = C.CDLL(os.path.join(refpipe.REFDIR, "nextcorrect.so"))
    rng = random.Random(5)
    for it in range(300):
        n1, n2 = rng.randint(0, 60), rng.randint(0, 60)
        if it < 5:
            n1, n2 = [(0, 0), (0, 7), (9, 0), (1, 1), (4, 4)][it]
        a = "".join(rng.choice("ACGT") for _ in range(n1))
        b = list(a)
        for _ in range(rng.randint(0, 8)):  # b = a with a few edits, or unrelated
            if b and rng.random() < 0.6:
                k = rng.randrange(len(b))
                b[k:k + 1] = rng.choice(["", "A", "CG", "T"])
        b = ("".join(b) if rng.random() < 0.8 else "".join(rng.choice("ACGT") for _ in range(n2)))[:60]
        out = []
        for lib in (native_lib, ref):
            al = util.Aln()
            tb, qb = C.create_string_buffer(len(a) + len(b) + 2), C.create_string_buffer(len(a) + len(b) + 2)
            al.t_aln_str, al.q_aln_str = C.cast(tb, C.c_char_p), C.cast(qb, C.c_char_p)
            lib.align_nd.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.POINTER(util.Aln)]
            lib.align_nd.restype = None
            lib.align_nd(a.encode(), len(a), b.encode(), len(b), C.byref(al))
            out.append((al.aln_len, tb.value, qb.value))
        assert out[0] == out[1], (a, b, out)


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


def test_overlap_exports_and_host_side():
    """libndgpu_overlap.so: every symbol of include/ndgpu_overlap.h is exported; the host-only entry points
    (preset table, .ovl encoder) work without a GPU, the device entry points refuse loudly."""
    import ctypes as C
    import numpy as np
    from nextdenovo_amd import overlap
    lib = overlap.load()
    header = os.path.join(os.path.dirname(util.HERE), "include", "ndgpu_overlap.h")
    names = _declared_functions(header)
    assert {"ndgpu_ovl_opt_preset", "ndgpu_ovl_index_create", "ndgpu_ovl_map", "ndgpu_ovl_encode", "ndgpu_ovl_sketch"} <= names
    for n in sorted(names):
        assert hasattr(lib, n), "missing export: " + n
    o = overlap.preset("ava-ont")
    assert (o.k, o.w, o.hpc, o.bw, o.max_gap, o.min_chain_score, o.minlen, o.no_diag, o.no_dual) == (15, 5, 0, 2000, 10000, 100, 500, 1, 1)
    o = overlap.preset("ava-pb")
    assert (o.k, o.w, o.hpc, o.bw) == (19, 5, 1, 500)
    # encoder: two records, delta / flag coding as lib/ovl.c:109-150
    recs = np.zeros(2, dtype=overlap.REC)
    recs[0] = (1, 300, 10, 2000, 5, 0, 1900, 777)
    recs[1] = (0, 200, 128, 700, 90, 16384, 17000, 3)
    prev = np.zeros(2, dtype=np.uint32)
    b = overlap.encode(recs, prev)
    from nextdenovo_amd import ovl
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".ovl") as f:
        f.write(b)
        f.flush()
        back = ovl.decode_ovl(f.name)
    assert back.tolist() == [[300, 1, 10, 2000, 5, 0, 1900, 777], [200, 0, 128, 700, 90, 16384, 17000, 3]]
    assert prev.tolist() == [200, 90]
    if lib.ndgpu_ovl_index_create and not _has_gpu():
        import pytest
        rs = overlap.ReadSet(np.zeros(1, np.uint32), np.asarray([40], np.uint32), np.zeros(3, np.uint32), np.zeros(1, np.uint64))
        with pytest.raises(RuntimeError):
            overlap.Index(o, rs)


def _has_gpu():
    from nextdenovo_amd import api
    try:
        return api.device_count() > 0
    except Exception:
        return False


def test_minimap2_nd_cli_host_logic():
    from nextdenovo_amd import minimap2_nd as m
    import numpy as np
    a = m.parse_argv("--step 1 --dual=yes -t 8 -x ava-ont -f 1000 -I 6G s.2bit p.2bit -o x.ovl".split())
    o = m.build_opt(a)
    assert (a.files, a.out, a.batch_size) == (["s.2bit", "p.2bit"], "x.ovl", 6000000000)
    assert (o.no_dual, o.mid_occ, o.minlen, o.k, o.w) == (0, 1000, 500, 15, 5)
    a = m.parse_argv("-x ava-pb --step 1 -f 0.0005 --minlen 1k a b".split())
    o = m.build_opt(a)
    assert (o.no_dual, o.mid_occ, o.minlen, o.hpc) == (1, 0, 1000, 1) and abs(o.mid_occ_frac - 0.0005) < 1e-9
    # --step is handled before the ordered options (main.c:185-200): its minlen default never overrides an explicit --minlen
    assert m.build_opt(m.parse_argv("-x ava-ont --minlen 1000 --step 1 a b".split())).minlen == 1000
    assert m.build_opt(m.parse_argv("--step 1 -x ava-ont a b".split())).minlen == 500
    assert m.parse_num("4G") == 4000000000 and m.parse_num("150k") == 150000 and m.parse_num("2.5m") == 2500000
    lens = np.asarray([60, 60, 60, 60, 60, 10], dtype=np.uint32)
    assert m.index_parts(lens, 100, mini_batch=50) == [(0, 2), (2, 4), (4, 6)]   # a part closes once its total exceeds -I
    assert m.index_parts(lens, 10**9) == [(0, 6)]
    import pytest
    a = m.parse_argv("--step 1 -x ava-hifi -f 800 -t 8 a b".split())   # the HiFi raw-align command (config_parser.py:46-47)
    o = m.build_opt(a)
    assert (o.k, o.w, o.hpc, o.mid_occ, o.bw, o.min_chain_score) == (51, 51, 1, 800, 500, 100)
    assert abs(m.build_opt(m.parse_argv("--step 1 -x ava-hifi a b".split())).mid_occ_frac - 1e-4) < 1e-9
    a = m.parse_argv("--step 1 -x ava-ont -c -z 300,150 -s 90 -O 5,20 -E 3 -A 3 -B 5 a b".split())   # -c and its scoring options
    o = m.build_opt(a)
    assert a.cigar and (a.aopt.zdrop, a.aopt.zdrop_inv, a.aopt.min_dp_max, a.aopt.q, a.aopt.q2, a.aopt.e, a.aopt.e2, a.aopt.a, a.aopt.b) == \
        (300, 150, 90, 5, 20, 3, 3, 3, 5)
    d = m.build_opt(m.parse_argv("--step 1 -x ava-pb -c a b".split()))
    assert (d.k, d.hpc) == (19, 1)
    for bad in ("--step 2 -x ava-ont a b", "--step 1 -x map-ont a b", "--step 1 -x ava-ont -a a b", "--step 2 -x ava-ont -c a b -o x",
                "--step 1 --mode 3 -x ava-ont -c a b"):
        with pytest.raises((SystemExit, ValueError)):
            m.build_opt(m.parse_argv(bad.split()))


def test_correct_stage_job_matrix():
    """raw_align's job list (reference nextDenovo:426-467): per seed file its part jobs, then the seed x seed jobs t >= i."""
    from nextdenovo_amd.correct_stage import job_matrix
    assert job_matrix(1, 1) == [(0, 0, "part", 0, True), (1, 0, "seed", 0, False)]
    assert job_matrix(2, 1) == [(0, 0, "part", 0, True), (1, 0, "seed", 0, False), (2, 0, "seed", 1, True),
                                (3, 1, "part", 0, True), (4, 1, "seed", 1, False)]
    assert [j[0] for j in job_matrix(3, 2)] == list(range(3 * 2 + 6))
