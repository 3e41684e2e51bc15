"""CPU-side checks of the product library: the C-ABI library loads and exports every symbol
include/kmc.h declares, the host logic (FASTA reader, generator, key decoding, owner function)
behaves, and there is no CPU fallback (kmc_create fails loudly without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, SAMPLE


def _has_gpu():
    return os.path.exists("/dev/kfd")


def test_library_exports_every_declared_symbol(kmc):
    hdr = open(os.path.join(ROOT, "include", "kmc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(kmc_[a-z_0-9]+)\s*\(", hdr)))
    assert declared == sorted(kmc.ABI_SYMBOLS)
    out = subprocess.run(["nm", "-D", "--defined-only", kmc.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (kmc_[a-z_0-9]+)", out))
    assert set(declared) <= exported
    L = kmc.lib()
    for s in declared:
        assert getattr(L, s) is not None
    assert b"gfx950" in L.kmc_version()


def test_no_torch_types_in_signatures():
    hdr = open(os.path.join(ROOT, "include", "kmc.h")).read()
    assert "torch" not in hdr and "at::" not in hdr and 'extern "C"' in hdr


@pytest.mark.skipif(_has_gpu(), reason="CPU-only check")
def test_create_fails_loudly_without_gpu(kmc):
    with pytest.raises(kmc.KmcError) as e:
        kmc.KmerCounter(k=31)
    assert e.value.status == kmc.ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_create_argument_checks(kmc):
    L = kmc.lib()
    h = C.c_void_p()
    assert L.kmc_create(C.byref(h), None) == kmc.ERR_ARG
    cfg = kmc._Config(3, 31, 0, 1, 0, 0, 0, None)  # wrong struct_size
    assert L.kmc_create(C.byref(h), C.byref(cfg)) == kmc.ERR_ARG
    cfg = kmc._Config(C.sizeof(kmc._Config), 64, 0, 1, 0, 0, 0, None)  # k out of range
    assert L.kmc_create(C.byref(h), C.byref(cfg)) == kmc.ERR_ARG
    assert b"1..63" in L.kmc_last_error(None)
    assert L.kmc_finalize(None, None, None) == kmc.ERR_ARG


def test_host_fasta_reader_matches_oracle(kmc, oracle, tmp_path):
    b1, o1 = kmc.parse_fasta(SAMPLE)
    b2, o2 = oracle.parse_fasta(SAMPLE)
    assert np.array_equal(b1, b2) and np.array_equal(o1, o2)
    assert o1.shape[0] == 201 and int(o1[-1]) == 80000
    cases = [b">r1 desc\nACGT  \r\nAC\n\n>r2\n>r3\nGG", b"", b">a\n", b">a\nAC\n>\n>b\nGG\n", b">x y z\nAAAA\nCCCC", b">a\r\nAC\r\n>b\r\nGT\r\n"]
    for i, data in enumerate(cases):
        p = tmp_path / f"c{i}.fasta"
        p.write_bytes(data)
        b1, o1 = kmc.parse_fasta(str(p))
        b2, o2 = oracle.parse_fasta(str(p))
        assert np.array_equal(b1, b2) and np.array_equal(o1, o2), data
    bad = tmp_path / "bad.fasta"
    bad.write_bytes(b"ACGT\n>r\nAC\n")
    with pytest.raises(kmc.KmcError) as e:
        kmc.parse_fasta(str(bad))
    assert e.value.status == kmc.ERR_FORMAT and "Expected > at record start." in str(e.value)
    with pytest.raises(kmc.KmcError) as e:
        kmc.parse_fasta(str(tmp_path / "nope.fasta"))
    assert e.value.status == kmc.ERR_IO


def test_host_fasta_reader_long_lines(kmc, oracle, tmp_path):
    rng = np.random.default_rng(5)
    seq = rng.integers(0, 4, 5_000_000)
    txt = np.frombuffer(b"ACGT", np.uint8)[seq].tobytes()
    p = tmp_path / "long.fasta"
    p.write_bytes(b">chr1\n" + txt[:4_500_000] + b"\n" + txt[4_500_000:] + b"\n>chr2\n" + txt[:100])
    b1, o1 = kmc.parse_fasta(str(p))
    b2, o2 = oracle.parse_fasta(str(p))
    assert np.array_equal(b1, b2) and np.array_equal(o1, o2) and o1.tolist() == [0, 5_000_000, 5_000_100]


def test_decode_key_and_owner(kmc):
    L = kmc.lib()
    buf = C.create_string_buffer(64)
    L.kmc_decode_key(0, 0b00011011, 4, buf)
    assert buf.raw[:4] == b"ACGT"
    L.kmc_decode_key(0b1101, 1 << 63, 34, buf)  # 68 bits: T, C from key_hi, then G, then zeros
    assert buf.raw[:34] == b"TCG" + b"A" * 31
    owners = [kmc.owner_of(0, i * 2654435761, 8) for i in range(4000)]
    assert set(owners) == set(range(8)) and all(kmc.owner_of(5, 7, 1) == 0 for _ in range(2))
    counts = np.bincount(owners, minlength=8)
    assert counts.min() > 400


def test_synth_generator_distribution(kmc, tmp_path):
    s = kmc.Synth(seed=7)
    bases, offs = kmc.synth_reads_host(s, 0, 300)
    assert offs.tolist() == [400 * i for i in range(301)]
    lines = bases.reshape(-1, 80)
    uniq = np.unique(lines, axis=0)
    assert uniq.shape[0] == 10  # pool of 10 lines, generator :5-8
    assert set(np.unique(bases).tolist()) == set(b"ACGT")
    # any record range is generated independently and identically
    b2, _ = kmc.synth_reads_host(s, 100, 50)
    assert np.array_equal(b2, bases[100 * 400:150 * 400])
    # a different seed gives a different pool
    b3, _ = kmc.synth_reads_host(kmc.Synth(seed=8), 0, 10)
    assert not np.array_equal(b3, bases[:4000])
    # pool=0: every line fresh random
    b4, _ = kmc.synth_reads_host(kmc.Synth(seed=7, pool=0), 0, 100)
    assert np.unique(b4.reshape(-1, 80), axis=0).shape[0] == 500
    # FASTA text: header format of generator :11-12, and it parses back to the same reads
    out = subprocess.run([os.path.join(ROOT, "bin", "kmc-genfasta"), "--records", "300", "--seed", "7"], capture_output=True, check=True).stdout
    assert out.startswith(b">dummy_sequence_001 1th record\n") and b">dummy_sequence_300 300th record\n" in out
    p = tmp_path / "g.fasta"
    p.write_bytes(out)
    pb, po = kmc.parse_fasta(str(p))
    assert np.array_equal(pb, bases) and np.array_equal(po, offs)
    # size parameter: smallest record count reaching N bytes, exact size reported
    for nbytes in (1, 436, 437, 87_492, 1_000_000, 12_345_678):
        n, exact = kmc.synth_records_for_bytes(s, nbytes)
        assert exact >= nbytes
        if nbytes <= 1_000_000:
            txt = subprocess.run([os.path.join(ROOT, "bin", "kmc-genfasta"), "--bytes", str(nbytes), "--seed", "7"], capture_output=True, check=True).stdout
            assert len(txt) == exact and txt.count(b">") == n
    # the reference's fixture shape: 200 records -> 87,492 bytes
    assert kmc.synth_records_for_bytes(s, 87_492) == (200, 87_492)


def test_table_formatting(kmc):
    t = kmc.Table(np.array([0, 0], np.uint64), np.array([0b0001, 0b1110], np.uint64), np.array([2, 1], np.uint64), 2)
    assert t.to_bytes() == b"AC\t2\nTG\t1\n"
    assert t.to_bytes(expand=True) == b"AC\nAC\nTG\n"
    assert t.n_total == 3


def test_host_fasta_reader_multi_segment(kmc, oracle, tmp_path):
    """Files above 32 MiB are parsed by several threads (16 MiB segments snapped to line starts):
    same result as the sequential oracle reader, including records and long lines that straddle
    segment boundaries, CRLF endings and a missing final newline."""
    exe = os.path.join(ROOT, "bin", "kmc-genfasta")
    p = tmp_path / "big.fasta"
    with open(p, "wb") as f:
        subprocess.run([exe, "--bytes", "60000000", "--seed", "11"], stdout=f, check=True)
    b1, o1 = kmc.parse_fasta(str(p))
    b2, o2 = oracle.parse_fasta(str(p))
    assert o1.shape[0] > 100_000 and np.array_equal(o1, o2) and np.array_equal(b1, b2)
    # one 20 MB line inside a 50 MB file + CRLF + no trailing newline
    rng = np.random.default_rng(3)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50_000_000)].tobytes()
    q = tmp_path / "long.fasta"
    q.write_bytes(b">a\r\n" + seq[:20_000_000] + b"\r\n" + seq[20_000_000:35_000_000] + b"\n>b x\n" + seq[35_000_000:] )
    b1, o1 = kmc.parse_fasta(str(q))
    b2, o2 = oracle.parse_fasta(str(q))
    assert o1.tolist() == [0, 35_000_000, 50_000_000] and np.array_equal(o1, o2) and np.array_equal(b1, b2)


def _stream_all(kmc, path, chunk_bytes):
    """The streaming reader's chunks concatenated into the whole-file form."""
    bs, os_, base, n_chunks = [], [np.zeros(1, np.uint64)], 0, 0
    for b, o in kmc.stream_fasta(path, chunk_bytes):
        assert int(o[0]) == 0 and int(o[-1]) == b.shape[0]
        bs.append(b)
        os_.append(o[1:] + np.uint64(base))
        base += int(b.shape[0])
        n_chunks += 1
    return (np.concatenate(bs) if bs else np.zeros(0, np.uint8)), np.concatenate(os_), n_chunks


def test_streaming_reader_matches_oracle(kmc, oracle, tmp_path):
    """kmc_fasta_stream_* (what kmc_count_file feeds the GPU from): every chunking of a file gives
    the whole-file reader's result -- chunks end at record starts, worker pieces are snapped to
    line starts, CRLF, blank lines, missing final newline, the empty-record terminator."""
    for cb in (1, 100, 437, 5000, 0):
        b1, o1, n = _stream_all(kmc, SAMPLE, cb)
        b2, o2 = oracle.parse_fasta(SAMPLE)
        assert np.array_equal(b1, b2) and np.array_equal(o1, o2), cb
        assert n == (200 if cb == 1 else n) and (cb != 0 or n == 1)
    cases = [b">r1 desc\nACGT  \r\nAC\n\n>r2\n>r3\nGG", b"", b">a\n", b">a\nAC\n>\n>b\nGG\n", b">x y z\nAAAA\nCCCC",
             b">a\r\nAC\r\n>b\r\nGT\r\n", b">a\nAC\n\n>b\n\nGG\n>\n\n>c\nTT\n", b">\n>b\nGG\n", b">a\nA>C\n>b\nG\n"]
    for i, data in enumerate(cases):
        p = tmp_path / f"s{i}.fasta"
        p.write_bytes(data)
        b2, o2 = oracle.parse_fasta(str(p))
        for cb in (1, 3, 7, 0):
            b1, o1, _ = _stream_all(kmc, str(p), cb)
            assert np.array_equal(b1, b2) and np.array_equal(o1, o2), (data, cb)
    bad = tmp_path / "sbad.fasta"
    bad.write_bytes(b"ACGT\n>r\nAC\n")
    with pytest.raises(kmc.KmcError) as e:
        list(kmc.stream_fasta(str(bad), 4))
    assert e.value.status == kmc.ERR_FORMAT and "Expected > at record start." in str(e.value)
    with pytest.raises(kmc.KmcError) as e:
        list(kmc.stream_fasta(str(tmp_path / "nope.fasta")))
    assert e.value.status == kmc.ERR_IO


def test_streaming_reader_multi_thread_chunks(kmc, oracle, tmp_path):
    """Chunks above 8 MiB are parsed by several threads (pieces of >= 4 MiB): generator-style text,
    then one 20 MB line, CRLF endings and no trailing newline."""
    exe = os.path.join(ROOT, "bin", "kmc-genfasta")
    p = tmp_path / "big.fasta"
    with open(p, "wb") as f:
        subprocess.run([exe, "--bytes", "60000000", "--seed", "12"], stdout=f, check=True)
    b2, o2 = oracle.parse_fasta(str(p))
    for cb in (0, 25_000_000, 9_000_000):
        b1, o1, n = _stream_all(kmc, str(p), cb)
        assert np.array_equal(o1, o2) and np.array_equal(b1, b2), cb
        assert n == (1 if cb == 0 else -(-60_000_000 // cb)) or n >= 2
    rng = np.random.default_rng(4)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 50_000_000)].tobytes()
    q = tmp_path / "long.fasta"
    q.write_bytes(b">a\r\n" + seq[:20_000_000] + b"\r\n" + seq[20_000_000:35_000_000] + b"\n>b x\n" + seq[35_000_000:])
    b2, o2 = oracle.parse_fasta(str(q))
    for cb in (0, 10_000_000):
        b1, o1, _ = _stream_all(kmc, str(q), cb)
        assert o1.tolist() == [0, 35_000_000, 50_000_000] and np.array_equal(o1, o2) and np.array_equal(b1, b2)


def _fastq_text(rng, n_rec, eol=b"\n", final_eol=True, max_len=300):
    """Random four-line FASTQ with N's and with quality strings that start with '@' or '+'."""
    recs, seqs = [], []
    for i in range(n_rec):
        L = int(rng.integers(0, max_len + 1))
        seq = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, L, p=[0.24, 0.24, 0.24, 0.24, 0.04])].tobytes()
        qual = np.frombuffer(b"@+!IIIIFFF#5", np.uint8)[rng.integers(0, 12, L)].tobytes()
        if L and i % 3 == 0:
            qual = b"@" + qual[1:]
        if L and i % 5 == 0:
            qual = b"+" + qual[1:]
        recs.append(b"@read%d some text" % i + eol + seq + eol + b"+" + (b"read%d" % i if i % 2 else b"") + eol + qual + eol)
        seqs.append(seq)
    text = b"".join(recs)
    if not final_eol and text.endswith(eol):
        text = text[:-len(eol)]
    offs = np.zeros(n_rec + 1, np.uint64)
    offs[1:] = np.cumsum([len(s) for s in seqs])
    return text, np.frombuffer(b"".join(seqs), np.uint8), offs


def test_streaming_reader_fastq(kmc, tmp_path):
    """Extension the reference does not have (parity unpinned): a file that starts
    with '@' is read as four-line FASTQ; only the sequence lines are handed on.  Every chunking, CRLF,
    missing final newline, quality lines that look like headers, multi-threaded chunks."""
    rng = np.random.default_rng(8)
    for eol, final_eol, n_rec in ((b"\n", True, 300), (b"\r\n", True, 120), (b"\n", False, 77), (b"\n", True, 1)):
        text, bases, offs = _fastq_text(rng, n_rec, eol, final_eol)
        p = tmp_path / "r.fastq"
        p.write_bytes(text)
        for cb in (1, 50, 997, 20000, 0):
            b1, o1, _ = _stream_all(kmc, str(p), cb)
            assert np.array_equal(o1, offs) and np.array_equal(b1, bases), (eol, final_eol, n_rec, cb)
    text, bases, offs = _fastq_text(rng, 120_000, max_len=250)   # ~32 MB: several worker pieces per chunk
    p = tmp_path / "big.fastq"
    p.write_bytes(text)
    for cb in (0, 9_000_000):
        b1, o1, n = _stream_all(kmc, str(p), cb)
        assert np.array_equal(o1, offs) and np.array_equal(b1, bases), cb
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@r1\nACGT\nIIII\n@r2\nAC\n+\nII\n")       # '+' line missing in the first record
    with pytest.raises(kmc.KmcError) as e:
        list(kmc.stream_fasta(str(bad)))
    assert e.value.status == kmc.ERR_FORMAT


def test_host_code_under_sanitizers(kmc, tmp_path):
    """kmc_host.cpp (readers, FASTQ, decode, generator) built with -fsanitize=address,undefined and
    run over edge-case files with several chunk sizes; the harness also checks streamed == whole-file."""
    exe = tmp_path / "host_reader_san"
    src = [os.path.join(ROOT, "tests", "native", "host_reader_san.cpp"), os.path.join(ROOT, "k-mer-count_amd", "csrc", "kmc_host.cpp")]
    r = subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                        "-I", os.path.join(ROOT, "include"), "-o", str(exe)] + src, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    cases = {"a.fasta": b">r1 desc\nACGT  \r\nAC\n\n>r2\n>r3\nGG", "b.fasta": b"", "c.fasta": b">a\n", "d.fasta": b">a\nAC\n>\n>b\nGG\n",
             "e.fasta": b"ACGT\n>r\nAC\n", "f.fasta": b">x\n" + b"ACGT" * 5000 + b"\n>y\n\n\n>z\nT", "g.fasta": b"\n", "h.fasta": b">",
             "i.fastq": b"@r1\nACGT\n+\nIIII\n@r2\nAC\n+r2\n@I\n", "j.fastq": b"@r1\nACGT\nIIII\n", "k.fastq": b"@", "l.fastq": b"@r\nAC\n+\nII"}
    files = []
    for name, data in cases.items():
        (tmp_path / name).write_bytes(data)
        files.append(str(tmp_path / name))
    files += [SAMPLE, str(tmp_path / "missing.fasta")]
    r = subprocess.run([str(exe)] + files, capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-1500:], r.stderr[-3000:])


# ---- the Rust binding (source only: no rustc in the build environment) checked against the header ----
_C2R = {"int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize", "double": "f64",
        "char": "c_char", "void": "c_void", "kmc_ctx": "KmcCtx", "kmc_config": "KmcConfig", "kmc_stats": "KmcStats",
        "kmc_reads": "KmcReads", "kmc_synth": "KmcSynth", "kmc_fasta_stream": "KmcFastaStream"}


def _c_type_to_rust(t):
    """'const uint64_t*' -> '*const u64', 'const void**' -> '*mut *const c_void', 'kmc_ctx**' -> '*mut *mut KmcCtx'."""
    t = t.strip()
    stars = t.count("*")
    base = t.replace("*", " ").split()
    const = "const" in base
    base = [b for b in base if b != "const"]
    assert len(base) == 1, t
    r = _C2R[base[0]]
    for i in range(stars):
        r = ("*const " if (const and i == 0) else "*mut ") + r
    return r


def _norm_rust(t):
    return re.sub(r"\s+", " ", t.strip()).replace("c_int", "i32")


def test_rust_binding_matches_the_header():
    """rust/src/lib.rs cannot be compiled here, so its extern "C" block and #[repr(C)] structs are parsed
    and compared with include/kmc.h: every declared function is bound, with the same number of
    arguments, the same argument and return types, and every struct has the same fields in the same order."""
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "kmc.h")).read(), flags=re.S)
    rs = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "src", "lib.rs")).read())
    # functions
    c_funcs = {}
    for ret, name, args in re.findall(r"^\s*([A-Za-z_][\w\s\*]*?)\s*\b(kmc_\w+)\s*\(([^)]*)\)\s*;", hdr, flags=re.M):
        args = [] if args.strip() in ("", "void") else [a.strip() for a in args.split(",")]
        c_funcs[name] = (ret.strip(), [re.sub(r"\s*\b\w+$", "", a) for a in args])
    block = re.search(r'extern "C" \{(.*?)\n\}', rs, flags=re.S).group(1)
    r_funcs = {}
    for name, args, ret in re.findall(r"pub fn (kmc_\w+)\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        args = [a.strip() for a in args.split(",") if a.strip()]
        r_funcs[name] = (_norm_rust(ret) if ret else "()", [_norm_rust(a.split(":", 1)[1]) for a in args])
    assert sorted(c_funcs) == sorted(r_funcs), (sorted(set(c_funcs) ^ set(r_funcs)))
    for name, (ret, args) in c_funcs.items():
        want_ret = "()" if ret == "void" else _c_type_to_rust(ret)
        assert r_funcs[name][0] == want_ret, (name, r_funcs[name][0], want_ret)
        assert r_funcs[name][1] == [_c_type_to_rust(a) for a in args], (name, r_funcs[name][1], args)
    # structs
    for cname, rname in (("kmc_config", "KmcConfig"), ("kmc_stats", "KmcStats"), ("kmc_reads", "KmcReads"), ("kmc_synth", "KmcSynth")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        c_fields = [(m[1], _c_type_to_rust(m[0])) for m in re.findall(r"([\w\s\*]+?)\s*\b(\w+)\s*;", body)]
        rbody = re.search(r"#\[repr\(C\)\]\s*pub struct %s \{(.*?)\}" % rname, rs, flags=re.S).group(1)
        r_fields = [(n, _norm_rust(t)) for n, t in re.findall(r"pub (\w+):\s*([^,]+),", rbody)]
        assert c_fields == r_fields, (cname, c_fields, r_fields)
    # constants the wrapper uses
    for cn, val in re.findall(r"^\s*(KMC_(?:MODE|ALGO)_\w+) = (\d+)", hdr, flags=re.M):
        m = re.search(r"pub const %s: i32 = (\d+);" % cn, rs)
        assert m and m.group(1) == val, cn


def test_host_entry_points_survive_allocation_failure(kmc, tmp_path):
    """kmc.h: "no exception or abort crosses the ABI".  In a child process the address space is capped
    (RLIMIT_AS) so that the host readers' allocations fail: every call must come back with a negative
    status (out of memory / I/O), the process must neither abort nor be killed, and the library must
    still work once the limit is lifted."""
    fa = tmp_path / "big.fasta"
    with open(fa, "wb") as f:
        subprocess.run([os.path.join(ROOT, "bin", "kmc-genfasta"), "--records", "600000", "--seed", "5"], stdout=f, check=True)
    assert os.path.getsize(fa) > 250_000_000
    child = r'''
import ctypes as C, resource, sys
L = C.CDLL(sys.argv[1])
class Reads(C.Structure):
    _fields_ = [("bases", C.c_void_p), ("offsets", C.c_void_p), ("n_reads", C.c_uint64), ("n_bases", C.c_uint64), ("max_read_len", C.c_uint64)]
L.kmc_parse_fasta.argtypes = [C.c_char_p, C.POINTER(Reads), C.c_char_p, C.c_size_t]
L.kmc_fasta_stream_open.argtypes = [C.c_char_p, C.c_uint64, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
L.kmc_fasta_stream_next.argtypes = [C.c_void_p, C.POINTER(Reads), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
L.kmc_fasta_stream_close.argtypes = [C.c_void_p]
L.kmc_free_reads.argtypes = [C.POINTER(Reads)]
path = sys.argv[2].encode()
eb = C.create_string_buffer(256)
def vm():
    for line in open("/proc/self/status"):
        if line.startswith("VmSize"):
            return int(line.split()[1]) * 1024
soft, hard = resource.getrlimit(resource.RLIMIT_AS)
results = []
for extra in (48 << 20, 96 << 20, 160 << 20):
    resource.setrlimit(resource.RLIMIT_AS, (vm() + extra, hard))
    rd = Reads()
    rc = L.kmc_parse_fasta(path, C.byref(rd), eb, 256)
    results.append(("parse", extra >> 20, rc))
    if rc == 0: L.kmc_free_reads(C.byref(rd))
    h = C.c_void_p()
    rc = L.kmc_fasta_stream_open(path, 64 << 20, C.byref(h), eb, 256)
    results.append(("open", extra >> 20, rc))
    if rc == 0:
        eof = C.c_int(0)
        while not eof.value:
            rc = L.kmc_fasta_stream_next(h, C.byref(rd), C.byref(eof), eb, 256)
            if rc: break
        results.append(("next", extra >> 20, rc))
        L.kmc_fasta_stream_close(h)
    resource.setrlimit(resource.RLIMIT_AS, (soft, hard))
rd = Reads()
rc = L.kmc_parse_fasta(path, C.byref(rd), eb, 256)
results.append(("parse_unlimited", 0, rc, rd.n_reads))
print(results)
'''
    import sys
    r = subprocess.run([sys.executable, "-c", child, kmc.LIB_PATH, str(fa)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout[-800:], r.stderr[-800:])   # no abort (SIGABRT = -6), no kill
    results = eval(r.stdout.strip().splitlines()[-1])
    failures = [x for x in results if x[0] != "parse_unlimited" and x[2] != 0]
    assert failures, results                                   # the cap did bite somewhere ...
    assert all(x[2] in (kmc.ERR_NOMEM, kmc.ERR_IO) for x in failures), results   # ... and came back as a status code
    assert results[-1][2] == 0 and results[-1][3] == 600_000   # and the library is intact afterwards


def test_cli_rejects_bad_numbers(kmc):
    """`-k abc` / `-k 0` must be an error (exit 2), not the reference's LR mode (atoi gave 0 = no -k)."""
    exe = os.path.join(ROOT, "bin", "k-mer-count")
    for argv in (["-k", "abc"], ["-k", "0"], ["-k", "64"], ["-k", "31x"], ["-k"], ["--gpus", "0"], ["--device", "-1"]):
        r = subprocess.run([exe, SAMPLE] + argv, capture_output=True, text=True)
        assert r.returncode == 2 and r.stdout == "" and "k-mer-count:" in r.stderr, (argv, r.returncode, r.stderr)
