#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ -- and, with --check, proves that the committed
ones are exactly what this script produces.

    python tests/golden/make_goldens.py           # (re)write lr_goldens.json, g1_expected.txt.gz, kat.json
    python tests/golden/make_goldens.py --check   # regenerate in memory, compare byte for byte, exit 1 on any difference

What is generated from what -- nothing below is produced by this repo's product code or by its C oracle:

* lr_goldens.json, g1_expected.txt.gz -- by EXECUTING the reference's own Python twin of main.rs,
  /root/reference/test.py (test.py:14-40), unmodified and in place (runpy; the file is never copied
  into the repo and nothing of it travels to the GPU box), on the reference's fixture
  k-mer-count/sample.fasta (G-full), on its first 6 / 18 lines (G-1 / G-3) and on a header-only file
  (G-empty), capturing stdout.  test.py's first statement is `from Bio import SeqIO` (test.py:2);
  Biopython is not installed in this image and cannot be installed, so the ONE thing test.py takes
  from it -- `SeqIO.parse(filename, "fasta")` yielding records whose `.seq` stringifies to the record's
  sequence lines joined (test.py:9-10) -- is supplied by an in-process stand-in injected into
  sys.modules (class _SeqIOStandIn below, 15 lines).  Consequence, stated where it matters
  (DESIGN.md section 2): the window loop, chunk-size loop, sort and output format of these goldens are
  the reference's own code; the FASTA *parsing* behind them is the stand-in's (strip + join of the
  lines between headers), so parser edge cases (blank lines, CRLF, ';' comments, lower case) stay
  "parity unpinned" -- every LR fixture is well-formed, LF-terminated, upper-case ACGT.
  G-1's full expected output (17,751 lines) is committed as bytes (gzip, mtime 0), so one fixture is
  the reference's output itself and not only its digest.
  This part needs /root/reference and is skipped (with a message) where that does not exist.
* kat.json -- known answers for CONTIGUOUS k (the reference has no k, no canonical
  strand and no count table, so these are definition-derived and "parity unpinned" by the reference)
  from an independent pure-Python probe (function kat_probe below: str slicing + dict counting on the
  committed copy of the fixture; it shares no code with oracle/kmc_oracle.c or the HIP kernels).
* sample.fasta -- the reference's fixture, a data file, committed byte for byte (sha256 checked here).
"""
import argparse
import contextlib
import gzip
import hashlib
import io
import json
import os
import runpy
import sys
import tempfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = "/root/reference"
REF_TEST_PY = os.path.join(REF_ROOT, "test.py")
REF_SAMPLE = os.path.join(REF_ROOT, "k-mer-count", "sample.fasta")
SAMPLE = os.path.join(HERE, "sample.fasta")
SAMPLE_SHA256 = "e4cccec3a83f90a380d18be975c1996460c86cdba014128b8f0b8db441ed1895"

LR_JSON = os.path.join(HERE, "lr_goldens.json")
KAT_JSON = os.path.join(HERE, "kat.json")
G1_GZ = os.path.join(HERE, "g1_expected.txt.gz")


# ---------------------------------------------------------------------------------------------
# the reference's test.py, executed in place
# ---------------------------------------------------------------------------------------------
class _Record:
    def __init__(self, seq):
        self.seq = seq


def _standin_parse(filename, fmt):
    """What test.py:9-10 relies on: one record per '>' header, .seq = its sequence lines stripped and joined."""
    assert fmt == "fasta"
    seq = None
    with open(filename, "r") as f:
        for line in f:
            if line.startswith(">"):
                if seq is not None:
                    yield _Record("".join(seq))
                seq = []
            elif seq is not None:
                seq.append(line.strip())
    if seq is not None:
        yield _Record("".join(seq))


def run_reference_test_py(fasta_path):
    """stdout (bytes) of `python /root/reference/test.py fasta_path`, run in this process."""
    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse = _standin_parse
    bio.SeqIO = seqio
    saved = {k: sys.modules.get(k) for k in ("Bio", "Bio.SeqIO")}
    saved_argv, saved_flag = sys.argv, sys.dont_write_bytecode
    sys.modules["Bio"], sys.modules["Bio.SeqIO"] = bio, seqio
    sys.argv = [REF_TEST_PY, fasta_path]
    sys.dont_write_bytecode = True  # never write into /root/reference
    buf = io.StringIO()
    try:
        with contextlib.redirect_stdout(buf):
            runpy.run_path(REF_TEST_PY, run_name="__main__")
    finally:
        sys.argv, sys.dont_write_bytecode = saved_argv, saved_flag
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return buf.getvalue().encode("ascii")


def _describe(out, with_extremes):
    """Digest + count-table view (`LC_ALL=C uniq -c`) of one expanded output."""
    lines = out.split(b"\n")
    assert lines[-1] == b""
    lines.pop()
    d = {"lines": len(lines), "bytes": len(out), "sha256": hashlib.sha256(out).hexdigest()}
    assert lines == sorted(lines)
    distinct, max_count, max_key = 0, 0, None
    uniq = hashlib.sha256()
    i, n = 0, len(lines)
    first_count = None
    while i < n:
        j = i
        while j < n and lines[j] == lines[i]:
            j += 1
        c = j - i
        if first_count is None:
            first_count = c
        if c > max_count:
            max_count, max_key = c, lines[i]
        uniq.update(b"%7d %s\n" % (c, lines[i]))
        distinct += 1
        i = j
    d["distinct"], d["max_count"] = distinct, max_count
    if with_extremes:
        d["max_key"] = max_key.decode()
        d["first_line"], d["last_line"] = lines[0].decode(), lines[-1].decode()
        d["first_count"] = first_count
        d["uniq_c_sha256"] = uniq.hexdigest()
    return d


def generate_lr():
    """(lr_goldens dict, G-1 expected output bytes) from the reference's test.py."""
    ref_sample = open(REF_SAMPLE, "rb").read()
    assert hashlib.sha256(ref_sample).hexdigest() == SAMPLE_SHA256, "reference fixture changed"
    assert open(SAMPLE, "rb").read() == ref_sample, "tests/golden/sample.fasta is not the reference's fixture"
    cases = {}
    g1_out = None
    with tempfile.TemporaryDirectory() as td:
        for name, head in (("G-full", None), ("G-1", 6), ("G-3", 18)):
            if head is None:
                path, extra = REF_SAMPLE, {"input_sha256": SAMPLE_SHA256}
            else:
                data = b"".join(ref_sample.splitlines(keepends=True)[:head])
                path = os.path.join(td, f"head{head}.fasta")
                open(path, "wb").write(data)
                extra = {"input_bytes": len(data), "input_sha256": hashlib.sha256(data).hexdigest()}
            out = run_reference_test_py(path)
            c = {"input": "sample.fasta", "head_lines": head}
            c.update(extra)
            c.update(_describe(out, with_extremes=head is None))
            cases[name] = c
            if name == "G-1":
                g1_out = out
        path = os.path.join(td, "empty.fasta")
        open(path, "wb").write(b">only_header\n")
        out = run_reference_test_py(path)
        cases["G-empty"] = {"input_text": ">only_header\n", "stdout_hex": out.hex(),
                            "note": "test.py:40 prints the join of an empty list: a lone newline (main.rs:35 panics instead)"}
    lr = {"source": "generated by tests/golden/make_goldens.py: stdout of /root/reference/test.py (unmodified, run in place; "
                    "Bio.SeqIO.parse supplied by the script's stand-in)",
          "cases": cases}
    return lr, g1_out


# ---------------------------------------------------------------------------------------------
# contiguous-k known answers: independent pure-Python probe of SURVEY.md 8a-def
# ---------------------------------------------------------------------------------------------
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def _records(path):
    recs, cur = [], None
    for line in open(path, "r"):
        if line.startswith(">"):
            if cur is not None:
                recs.append("".join(cur))
            cur = []
        else:
            cur.append(line.rstrip())
    if cur is not None:
        recs.append("".join(cur))
    return recs


def kat_probe(path, k, canonical):
    """{kmer: count} by the definitions of 8a-def: windows of k inside one record, stride 1; a window
    with a character outside ACGT is skipped; canonical = min(kmer, reverse complement) as strings
    (A<C<G<T is ASCII order, main.rs:19-22,87)."""
    table = {}
    for seq in _records(path):
        for i in range(len(seq) - k + 1):
            w = seq[i:i + k]
            if any(ch not in _COMP for ch in w):
                continue
            if canonical:
                rc = "".join(_COMP[ch] for ch in reversed(w))
                if rc < w:
                    w = rc
            table[w] = table.get(w, 0) + 1
    return table


def generate_kat():
    cases = {}
    for k in (5, 21, 31, 63):
        c = {}
        for canonical, tag in ((False, "fwd"), (True, "canon")):
            t = kat_probe(SAMPLE, k, canonical)
            keys = sorted(t)
            text = "".join("%s\t%d\n" % (key, t[key]) for key in keys).encode()
            c["total"] = sum(t.values())
            c[f"distinct_{tag}"] = len(t)
            c[f"max_{tag}"] = max(t.values())
            c[f"digest_{tag}"] = hashlib.sha256(text).hexdigest()[:16]
            c[f"sha256_{tag}"] = hashlib.sha256(text).hexdigest()
            if canonical:
                top = max(t.values())
                c["top_canon"] = min(key for key in keys if t[key] == top)  # first in sorted order among the most frequent
        cases[str(k)] = c
    return {"source": "generated by tests/golden/make_goldens.py:kat_probe (pure Python, SURVEY.md 8a-def) on tests/golden/sample.fasta; "
                      "digest = first 16 hex of sha256 over 'KMER\\tCOUNT\\n' lines sorted by KMER; parity unpinned by the reference (it has no k)",
            "cases": cases}


# ---------------------------------------------------------------------------------------------
def _json_bytes(obj):
    return (json.dumps(obj, indent=1, sort_keys=True) + "\n").encode()


def _gz_bytes(data):
    buf = io.BytesIO()
    with gzip.GzipFile(filename="", mode="wb", fileobj=buf, compresslevel=9, mtime=0) as f:
        f.write(data)
    return buf.getvalue()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true", help="regenerate in memory and compare with the committed files")
    args = ap.parse_args()
    assert hashlib.sha256(open(SAMPLE, "rb").read()).hexdigest() == SAMPLE_SHA256
    products = {KAT_JSON: _json_bytes(generate_kat())}
    if os.path.exists(REF_TEST_PY):
        lr, g1 = generate_lr()
        products[LR_JSON] = _json_bytes(lr)
        products[G1_GZ] = _gz_bytes(g1)
    else:
        print(f"note: {REF_TEST_PY} not present -- LR goldens not regenerated (they need the reference)")
    bad = 0
    for path, data in products.items():
        if args.check:
            have = open(path, "rb").read() if os.path.exists(path) else None
            same = have == data
            if not same and path.endswith(".gz") and have is not None:
                same = gzip.decompress(have) == gzip.decompress(data)  # (zlib versions may differ in the stream)
            print(("ok       " if same else "DIFFERS  ") + os.path.relpath(path, HERE))
            bad += 0 if same else 1
        else:
            open(path, "wb").write(data)
            print("wrote", os.path.relpath(path, HERE), len(data), "bytes")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
