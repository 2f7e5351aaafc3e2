"""The parallel gzip inflater of the CLI's ingest (bcalm_amd/host/pgz.h; SURVEY.md section 8 row f2, /root/reference/README.md:45-50:
gzipped FASTA / FASTQ input): `pgz_cat` (the same header as a filter) against zlib byte for byte, and the simulator build of the
`bcalm` CLI on gzip files that are cut into many chunks."""
import gzip
import json
import os
import random
import struct
import subprocess
import zlib

import pytest

import oracle_lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "bcalm_amd", "_build", "pgz_cat")


@pytest.fixture(scope="module")
def pgz_cat():
    import __graft_entry__
    __graft_entry__.build_host_tools()
    assert os.path.exists(EXE)
    return EXE


def _fastq(rng, n, L=150, wrap=False):
    g = "".join(rng.choice("ACGT") for _ in range(50000))
    out = []
    for i in range(n):
        s = rng.randrange(0, len(g) - L)
        r = "".join(rng.choice("ACGT") if rng.random() < 0.01 else c for c in g[s:s + L])
        q = "".join(chr(33 + min(40, max(2, int(rng.gauss(30, 7))))) for _ in range(L))
        out.append("@SRR1.%d %d/1 length=%d\n%s\n+\n%s\n" % (i, i, L, r, q))
    return "".join(out).encode()


def _run(exe, path, threads, chunk):
    r = subprocess.run([exe, str(path), str(threads), str(chunk)], capture_output=True)
    return r.returncode, r.stdout, r.stderr.decode()


def _deflate(data, level=6, wbits=31, mem=8, strategy=zlib.Z_DEFAULT_STRATEGY, flush_every=0, flush=zlib.Z_SYNC_FLUSH):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits, mem, strategy)
    if not flush_every:
        return c.compress(data) + c.flush()
    out = b""
    for i in range(0, len(data), flush_every):
        out += c.compress(data[i:i + flush_every]) + c.flush(flush)
    return out + c.flush()


@pytest.fixture(scope="module")
def text():
    return _fastq(random.Random(7), 12000)          # 4 MB


def test_levels_strategies_members(pgz_cat, text, tmp_path):
    cases = {
        "l1": _deflate(text, 1), "l6": _deflate(text, 6), "l9": _deflate(text, 9),
        "huffman_only": _deflate(text, 6, strategy=zlib.Z_HUFFMAN_ONLY), "rle": _deflate(text, 6, strategy=zlib.Z_RLE), "filtered": _deflate(text, 6, strategy=zlib.Z_FILTERED),
        "window_512": _deflate(text, 6, wbits=16 + 9), "window_4k": _deflate(text, 9, wbits=16 + 12),
        "small_blocks": _deflate(text, 6, mem=3), "sync_flushes": _deflate(text, 6, flush_every=70001), "full_flushes": _deflate(text, 6, flush_every=50021, flush=zlib.Z_FULL_FLUSH),
        "members": b"".join(gzip.compress(text[i:i + 300007], 6) for i in range(0, len(text), 300007)),
        "bgzf_sized_members": b"".join(gzip.compress(text[i:i + 65280], 6) for i in range(0, len(text), 65280)),
        "members_zero_padded": gzip.compress(text[:1000000]) + b"\0" * 37 + gzip.compress(text[1000000:]),
        "trailing_garbage": _deflate(text, 6) + b"not a gzip member at all",
        "bgzf_with_its_eof_marker": b"".join(gzip.compress(text[i:i + 65280], 6) for i in range(0, len(text), 65280)) + bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000"),
    }
    for name, blob in cases.items():
        p = tmp_path / (name + ".gz"); p.write_bytes(blob)
        for threads, chunk in ((4, 100000), (3, 37000), (8, 400000)):
            rc, out, err = _run(pgz_cat, p, threads, chunk)
            assert rc == 0, (name, threads, chunk, err)
            assert out == text, (name, threads, chunk, err)
            st = json.loads(err)
            assert st["out_bytes"] == len(text) and st["chunks_on_chain"] >= 2, (name, err)


def test_header_fields(pgz_cat, text, tmp_path):
    """FEXTRA, FNAME, FCOMMENT, FHCRC in the member header (RFC 1952)"""
    raw = zlib.compressobj(6, zlib.DEFLATED, -15); body = raw.compress(text) + raw.flush()
    hdr = b"\x1f\x8b\x08" + bytes([4 | 8 | 16 | 2]) + b"\0\0\0\0\x00\x03" + struct.pack("<H", 6) + b"BC\x02\x00\x11\x22" + b"reads.fastq\0" + b"a comment\0"
    hdr += struct.pack("<H", zlib.crc32(hdr) & 0xFFFF)
    blob = hdr + body + struct.pack("<II", zlib.crc32(text), len(text) & 0xFFFFFFFF)
    assert gzip.decompress(blob) == text
    p = tmp_path / "h.gz"; p.write_bytes(blob)
    rc, out, err = _run(pgz_cat, p, 4, 100000)
    assert rc == 0 and out == text, err


def test_not_handled_and_errors(pgz_cat, text, tmp_path):
    """stored / fixed-code streams, binary data, tiny files: 'not handled' (exit 2; the CLI then inflates with zlib) and never wrong bytes;
    a damaged stream is an error or 'not handled', never a success"""
    rng = random.Random(3)
    for name, blob in {"stored": gzip.compress(text, 0), "fixed": _deflate(text, 6, strategy=zlib.Z_FIXED), "binary": gzip.compress(bytes(rng.randrange(256) for _ in range(300000)), 6),
                       "tiny": gzip.compress(text[:5000], 6), "not_gzip": text[:200000]}.items():
        p = tmp_path / (name + ".gz"); p.write_bytes(blob)
        rc, out, err = _run(pgz_cat, p, 4, 50000)
        assert rc == 2 and out == b"", (name, rc, err)
    good = _deflate(text, 6)
    bad_crc = good[:-8] + struct.pack("<I", (zlib.crc32(text) ^ 1) & 0xFFFFFFFF) + good[-4:]
    bad_len = good[:-4] + struct.pack("<I", (len(text) + 1) & 0xFFFFFFFF)
    truncated = good[:len(good) * 2 // 3]
    flipped = bytearray(good); flipped[len(good) // 2] ^= 0x10
    for name, blob in {"bad_crc": bad_crc, "bad_len": bad_len, "truncated": truncated, "flipped_bit": bytes(flipped)}.items():
        p = tmp_path / (name + ".gz"); p.write_bytes(blob)
        for threads, chunk in ((2, 100000), (8, 100000)):          # (several waves: text has been handed over when the damage shows; one wave: nothing has)
            rc, out, err = _run(pgz_cat, p, threads, chunk)
            assert rc in (1, 2), (name, rc, err)
            assert text.startswith(out) or name == "flipped_bit", name


def test_random_texts(pgz_cat, tmp_path):
    """FASTA with long lines, low-complexity text, short records; random deflate parameters"""
    rng = random.Random(11)
    for it in range(12):
        kind = it % 3
        if kind == 0:
            t = "".join(">chr%d\n%s\n" % (i, "\n".join("".join(rng.choice("ACGT") for _ in range(70)) for _ in range(rng.randrange(1, 400)))) for i in range(40)).encode()
        elif kind == 1:
            t = ("@r\n" + "ACGT" * 30 + "\n+\n" + "I" * 120 + "\n").encode() * rng.randrange(20000, 40000)
        else:
            t = _fastq(rng, 6000, L=rng.choice([36, 75, 250]))
        blob = _deflate(t, rng.choice([1, 4, 6, 9]), wbits=16 + rng.choice([9, 12, 15]), mem=rng.choice([2, 4, 8, 9]), strategy=rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE]),
                        flush_every=rng.choice([0, 0, 30011]))
        p = tmp_path / ("r%d.gz" % it); p.write_bytes(blob)
        rc, out, err = _run(pgz_cat, p, rng.choice([2, 5, 8]), rng.choice([20000, 64000, 250000]))
        assert rc in (0, 2), (it, err)
        if rc == 0:
            assert out == t, (it, err)
        else:
            assert out == b""


# ---- the CLI (simulator build): a gzip file cut into many chunks gives the graph of the plain text ----
@pytest.fixture(scope="module")
def cli():
    import hostsim_lib
    hostsim_lib.load()
    exe = os.path.join(ROOT, "tests", "hostsim", "_build", "bcalm_hostsim")
    assert os.path.exists(exe)
    return exe


@pytest.fixture(scope="module")
def oracle():
    return oracle_lib.load()


def _cli_run(cli, tmp_path, args, env):
    e = dict(os.environ); e.update(env)
    r = subprocess.run([cli] + args + ["-kmer-size", "21", "-abundance-min", "1", "-out", "o"], cwd=tmp_path, capture_output=True, text=True, env=e)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = open(tmp_path / "o.unitigs.fa").read().split("\n")
    recs = [(lines[i + 1], int(lines[i].split("KC:i:")[1].split()[0])) for i in range(0, len(lines) - 1, 2)]
    return recs, r.stdout + r.stderr


def test_cli_gzip_in_chunks(cli, oracle, tmp_path):
    rng = random.Random(5)
    g = "".join(rng.choice("ACGT") for _ in range(3000))
    reads = []
    for _ in range(700):
        L = rng.randrange(60, 200); s = rng.randrange(0, len(g) - L); reads.append(g[s:s + L])
    exp = oracle.run("\n".join(reads) + "\n", 21, 1)
    nb = sum(len(r) for r in reads)
    fq = "".join("@read%d\n%s\n+\n%s\n" % (i, r, ("@+I" * len(r))[:len(r)]) for i, r in enumerate(reads)).encode()
    fa = "".join(">read%d\n%s\n" % (i, "\n".join(r[j:j + 50] for j in range(0, len(r), 50))) for i, r in enumerate(reads)).encode()
    env = {"BCALM_GZ_CHUNK": "3000", "BCALM_GZ_VERBOSE": "1", "CDBG_STAGE_BYTES": "8192"}
    for name, t in (("fq", fq), ("fa", fa)):
        (tmp_path / (name + ".gz")).write_bytes(_deflate(t, 6, mem=4))          # (blocks of 1024 symbols: a few dozen block starts in a small file)
        recs, out = _cli_run(cli, tmp_path, ["-in", name + ".gz", "-nb-cores", "4"], env)
        assert oracle_lib.canonical_set(oracle, recs, 21) == exp["unitigs"], name
        assert "input: 700 sequences, %d bases" % nb in out and "inflated by 4 threads" in out, out
    # several members; one thread (zlib path); BCALM_GZ_SERIAL
    (tmp_path / "m.gz").write_bytes(b"".join(_deflate(fq[i:i + 40000], 6, mem=4) for i in range(0, len(fq), 40000)))
    recs, out = _cli_run(cli, tmp_path, ["-in", "m.gz", "-nb-cores", "3"], env)
    assert oracle_lib.canonical_set(oracle, recs, 21) == exp["unitigs"] and "inflated by 3 threads" in out
    for args, e in ((["-in", "fq.gz", "-nb-cores", "1"], env), (["-in", "fq.gz", "-nb-cores", "4"], dict(env, BCALM_GZ_SERIAL="1"))):
        recs, out = _cli_run(cli, tmp_path, args, e)
        assert oracle_lib.canonical_set(oracle, recs, 21) == exp["unitigs"] and "inflated by" not in out
    # a FASTQ whose records wrap: the first wave says so before anything is pushed -> zlib + the tolerant parser
    wrapped = "".join("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:30], r[30:], "I" * 30, "I" * (len(r) - 30)) for i, r in enumerate(reads)).encode()
    (tmp_path / "w.gz").write_bytes(_deflate(wrapped, 6, mem=4))
    recs, out = _cli_run(cli, tmp_path, ["-in", "w.gz", "-nb-cores", "4"], env)
    assert oracle_lib.canonical_set(oracle, recs, 21) == exp["unitigs"] and "input: 700 sequences" in out and "inflated by" not in out
    # ... and one whose FIRST record is four lines but a later one wraps: found after text was pushed -> the whole ingest starts over
    mixed = "".join(("@r%d\n%s\n%s\n+\n%s\n%s\n" % (i, r[:30], r[30:], "I" * 30, "@" * (len(r) - 30))) if i == 400 else ("@r%d\n%s\n+\n%s\n" % (i, r, "I" * len(r))) for i, r in enumerate(reads)).encode()
    (tmp_path / "x.gz").write_bytes(_deflate(mixed, 6, mem=4))
    recs, out = _cli_run(cli, tmp_path, ["-in", "x.gz", "-nb-cores", "4"], env)
    assert oracle_lib.canonical_set(oracle, recs, 21) == exp["unitigs"] and "input: 700 sequences" in out
    # a damaged file is an error, not a shorter graph
    blob = bytearray(_deflate(fq, 6, mem=4)); blob[len(blob) // 2] ^= 0x04
    (tmp_path / "bad.gz").write_bytes(bytes(blob))
    for cores in ("4", "1"):
        e = dict(os.environ); e.update(env)
        r = subprocess.run([cli, "-in", "bad.gz", "-kmer-size", "21", "-abundance-min", "1", "-nb-cores", cores], cwd=tmp_path, capture_output=True, text=True, env=e)
        assert r.returncode != 0, r.stdout
