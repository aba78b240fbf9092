"""Native BLOW5 writer (sqg_blow5_*, squigulator_amd/csrc/h_blow5.h) against the files the compiled reference wrote through
its own slow5lib (tests/golden/blow5, tools/make_blow5_golden.py): `cmp`-identical, header to "5WOLB".

CPU: the writer is host code; the svb-zd bytes it frames come from the oracle's coder here (itself pinned on slow5lib's
bytes, tests/test_svb.py).  GPU: the whole product path -- reads from the fixture, signals from the kernels, svb-zd on the
device, sqg_blow5_write_batch -- must produce the same file, and a full-size batch must decode back to its signals."""
import os
import struct
import zlib

import numpy as np
import pytest

import orc
from blow5_cases import BLOW5_CASES
from squigulator_amd import api, model, options, profiles

import subprocess

GOLD = os.path.join(os.path.dirname(__file__), "golden", "blow5")
DUMP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_blow5_dump")   # the reference's slow5lib as a reader
INPUTS = os.path.join(os.path.dirname(__file__), "golden", "inputs")


def _case(cid):
    d = np.load(os.path.join(GOLD, cid + ".npz"))
    o = options.parse_args(str(d["cmd"]))
    ids = bytes(d["ids"]).split(b"\n")
    so = np.concatenate(([0], np.cumsum(d["lens"]))).astype(np.int64)
    return o, ids, d["offset"], d["median"], so, d["sig"]


def parse_blow5(buf):
    """-> (header text, [uncompressed records]); checks magic, sizes and the EOF marker"""
    assert buf[:6] == b"BLOW5\x01" and buf[6:9] == b"\x00\x02\x00" and buf[9] == 1 and buf[14] == 1
    assert struct.unpack_from("<I", buf, 10)[0] == 1 and buf[15:64] == b"\0" * 49
    hs = struct.unpack_from("<I", buf, 64)[0]
    p = 68 + hs
    recs = []
    while buf[p:] != b"5WOLB":
        (n,) = struct.unpack_from("<Q", buf, p)
        recs.append(zlib.decompress(buf[p + 8:p + 8 + n]))
        p += 8 + n
    return buf[68:68 + hs], recs


@pytest.mark.parametrize("cid", [c[0] for c in BLOW5_CASES])
@pytest.mark.parametrize("threads", [1, 3])
def test_writer_reproduces_the_reference_file(cid, threads, tmp_path):
    o, ids, offset, median, so, sig = _case(cid)
    want = open(os.path.join(GOLD, cid + ".blow5"), "rb").read()
    encs = [orc.svb_zd(sig[so[i]:so[i + 1]]) for i in range(len(ids))]
    path = str(tmp_path / "x.blow5")
    w = api.Blow5Writer(path, o.profile, o.flags, threads=threads)
    # in the reference's batches (-K): read_number and start_time carry over the calls
    done = 0
    while done < len(ids):
        nb = min(o.batch, len(ids) - done)
        e = encs[done:done + nb]
        eo = np.concatenate(([0], np.cumsum([len(x) for x in e]))).astype(np.int64)
        w.write(ids[done:done + nb], offset[done:done + nb], median[done:done + nb], so[done:done + nb + 1] - so[done],
                np.concatenate(e), eo)
        done += nb
    n = w.close()
    got = open(path, "rb").read()
    assert n == len(got)
    hdr_g, rec_g = parse_blow5(got)
    hdr_w, rec_w = parse_blow5(want)
    assert hdr_g == hdr_w
    assert rec_g == rec_w                      # the uncompressed records (independent of the zlib build)
    assert got == want                         # and the very bytes (same zlib as the reference build in this image)


def test_empty_file_and_bad_arguments(tmp_path):
    prof, fl = profiles.get_profile("dna-r9-prom")
    w = api.Blow5Writer(str(tmp_path / "e.blow5"), prof, fl)
    w.write([], np.zeros(0), np.zeros(0), np.zeros(1, np.int64), np.zeros(1, np.uint8), np.zeros(1, np.int64))
    assert w.close() == os.path.getsize(tmp_path / "e.blow5")
    hdr, recs = parse_blow5(open(tmp_path / "e.blow5", "rb").read())
    assert recs == [] and b"@sequencing_kit\tsqk-lsk109\n" in hdr
    with pytest.raises(api.SqgError):
        api.Blow5Writer(str(tmp_path / "no" / "such" / "dir.blow5"), prof, fl)


@pytest.mark.gpu
@pytest.mark.parametrize("cid", ["r9_t1", "r10_t1", "rna004_prefix", "r9_two_batches"])
def test_product_path_writes_the_reference_file(cid, tmp_path):
    """fixture reads -> kernels -> svb-zd on the device -> sqg_blow5_write_batch == the reference's BLOW5"""
    o, ids, offset, median, so, sig = _case(cid)
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "refvec", "r9_t1.npz"))     # (layout of the refvec fixtures)
    del d
    # the reads themselves: sample them as the reference did (device sampler on the same FASTA, same seed)
    import bench
    contigs = bench.load_contigs(os.path.join(INPUTS, o.ref))
    k = o.kmer_size_default
    mean, stdv = model.synthetic_model(k)
    gen = api.SignalGenerator(o.profile, o.flags, k, mean, stdv, o.seed, num_workers=o.threads, mode=api.MODE_CERTIFIED)
    gen.load_genome(contigs, o.rlen, api.SAMPLE_RNA if (o.flags & profiles.SQ_RNA) else api.SAMPLE_DNA)
    path = str(tmp_path / "x.blow5")
    w = api.Blow5Writer(path, o.profile, o.flags, threads=2)
    done = 0
    while done < len(ids):
        nb = min(o.batch, len(ids) - done)
        b = gen.sample(nb).run().wait()
        np.testing.assert_array_equal(b.offset, offset[done:done + nb])
        w.write_batch(b, ids[done:done + nb])
        b.free()
        done += nb
    w.close()
    gen.close()
    assert open(path, "rb").read() == open(os.path.join(GOLD, cid + ".blow5"), "rb").read()


@pytest.mark.gpu
def test_full_size_batch_round_trip(tmp_path):
    """a bench-sized batch: every record inflates, its signal field decodes (svb-zd) to the batch's int16 samples"""
    import bench
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    gen.load_genome(bench.synthetic_genome_host(8.0), 10000, api.SAMPLE_DNA)
    b = gen.sample(1024).run().wait()
    sig = b.signal()
    ids = [b"S1_%d!c!0!1!+" % (i + 1) for i in range(b.n_reads)]
    path = str(tmp_path / "big.blow5")
    w = api.Blow5Writer(path, prof, fl)
    w.write_batch(b, ids)
    n = w.close()
    buf = open(path, "rb").read()
    assert n == len(buf) and n < 2 * len(sig) * 0.8              # smaller than the raw int16
    _, recs = parse_blow5(buf)
    assert len(recs) == b.n_reads
    start = 0
    for i in (0, 1, 17, 500, b.n_reads - 1):
        r = recs[i]
        (idl,) = struct.unpack_from("<H", r, 0)
        q = 2 + idl + 4
        dig, off, rng_, sr = struct.unpack_from("<4d", r, q); q += 32
        (nb,) = struct.unpack_from("<Q", r, q); q += 8
        dec, used = orc.svb_zd_decode(np.frombuffer(r, np.uint8, nb, q))
        np.testing.assert_array_equal(dec, sig[b.sig_off[i]:b.sig_off[i + 1]])
        assert used == nb and off == b.offset[i] and dig == prof.digitisation
        q += nb
        one, ch = struct.unpack_from("<Qc", r, q); q += 9
        med, rn, mux, st = struct.unpack_from("<diBQ", r, q)
        assert (one, ch, med, rn, mux, st) == (1, b"0", b.median_before[i], i, 0, int(b.sig_off[i]))
    b.free(); gen.close()


# ---- SQG_BLOW5_STORED: the same records in zlib streams of stored blocks (include/sqg.h) ------------------------------------------
def ref_dump(path, how="full"):
    """the file as the REFERENCE's own slow5lib reads it (oracle/_ref/ref_blow5_dump: header attributes, every field of every record,
    every sample): None where the reader was not built (no /root/reference at build time)"""
    if not os.path.exists(DUMP):
        return None
    return subprocess.run([DUMP, path] + (["hash"] if how == "hash" else []), capture_output=True, text=True, check=True).stdout


def _stored_records_are_valid(buf):
    """every record's stream is 78 01 + stored blocks + Adler-32, nothing else: sizes as include/sqg.h states them"""
    hs = struct.unpack_from("<I", buf, 64)[0]
    p = 68 + hs
    while buf[p:] != b"5WOLB":
        (n,) = struct.unpack_from("<Q", buf, p)
        z = buf[p + 8:p + 8 + n]
        assert z[:2] == b"\x78\x01"
        q, raw = 2, b""
        while True:
            final, ln, nln = z[q], *struct.unpack_from("<HH", z, q + 1)
            assert final in (0, 1) and ln == (~nln & 0xffff) and (final == 1 or ln == 65535)
            raw += z[q + 5:q + 5 + ln]
            q += 5 + ln
            if final:
                break
        assert q + 4 == n and struct.unpack_from(">I", z, q)[0] == zlib.adler32(raw)
        p += 8 + n


@pytest.mark.parametrize("cid", [c[0] for c in BLOW5_CASES])
def test_stored_mode_holds_the_reference_records(cid, tmp_path):
    """host framing (sqg_blow5_write with SQG_BLOW5_STORED): same header, same uncompressed records as the reference's file, and the
    reference's own slow5lib reads both files to the same text, field for field and sample for sample"""
    o, ids, offset, median, so, sig = _case(cid)
    gold = os.path.join(GOLD, cid + ".blow5")
    encs = [orc.svb_zd(sig[so[i]:so[i + 1]]) for i in range(len(ids))]
    path = str(tmp_path / "s.blow5")
    w = api.Blow5Writer(path, o.profile, o.flags, threads=3, stored=True)
    done = 0
    while done < len(ids):
        nb = min(o.batch, len(ids) - done)
        e = encs[done:done + nb]
        eo = np.concatenate(([0], np.cumsum([len(x) for x in e]))).astype(np.int64)
        w.write(ids[done:done + nb], offset[done:done + nb], median[done:done + nb], so[done:done + nb + 1] - so[done], np.concatenate(e), eo)
        done += nb
    n = w.close()
    got = open(path, "rb").read()
    assert n == len(got)
    hdr_g, rec_g = parse_blow5(got)
    hdr_w, rec_w = parse_blow5(open(gold, "rb").read())
    assert hdr_g == hdr_w and rec_g == rec_w
    _stored_records_are_valid(got)
    d = ref_dump(path)
    if d is not None:
        assert d == ref_dump(gold) and d.strip().endswith("records\t%d" % len(ids))


def test_stored_mode_long_records(tmp_path):
    """records of several stored blocks (> 65535 bytes), of exactly one full block, and ids of every length: sizes and checksums"""
    prof, fl = profiles.get_profile("dna-r9-prom")
    rng = np.random.default_rng(3)
    lens = [1, 30000, 65535 - 107, 65535 - 106, 65535 - 105, 200000, 131070 - 106]
    sig = [rng.integers(300, 900, m).astype(np.int16) for m in lens]
    encs = [orc.svb_zd(x) for x in sig]
    ids = [b"r%d" % i + b"x" * (i * 7) for i in range(len(lens))]
    so = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
    eo = np.concatenate(([0], np.cumsum([len(x) for x in encs]))).astype(np.int64)
    path = str(tmp_path / "l.blow5")
    w = api.Blow5Writer(path, prof, fl, threads=2, stored=True)
    w.write(ids, rng.normal(10, 3, len(lens)), rng.normal(200, 20, len(lens)), so, np.concatenate(encs), eo)
    w.close()
    buf = open(path, "rb").read()
    _stored_records_are_valid(buf)
    _, recs = parse_blow5(buf)
    for i, r in enumerate(recs):
        q = 2 + len(ids[i]) + 4 + 32
        (nb,) = struct.unpack_from("<Q", r, q)
        dec, used = orc.svb_zd_decode(np.frombuffer(r, np.uint8, nb, q + 8))
        np.testing.assert_array_equal(dec, sig[i])
    d = ref_dump(path, "hash")
    assert d is None or d.strip().endswith("records\t%d" % len(lens))


@pytest.mark.gpu
@pytest.mark.parametrize("cid", ["r9_t1", "r10_t1", "rna004_prefix", "r9_two_batches", "r9_ont"])
def test_device_framed_records_equal_the_host_framing_and_the_reference_records(cid, tmp_path):
    """the product path in stored mode: fixture reads -> kernels -> svb-zd -> records framed on the device (k_blow5_frame) -> file.  The
    bytes are those of the host framing of the same records (sqg_blow5_write with the flag), the uncompressed records the reference
    file's, and the reference's slow5lib reads both files to the same text"""
    o, ids, offset, median, so, sig = _case(cid)
    import bench
    contigs = bench.load_contigs(os.path.join(INPUTS, o.ref))
    k = o.kmer_size_default
    mean, stdv = model.synthetic_model(k)
    gen = api.SignalGenerator(o.profile, o.flags, k, mean, stdv, o.seed, num_workers=o.threads, mode=api.MODE_CERTIFIED)
    gen.load_genome(contigs, o.rlen, api.SAMPLE_RNA if (o.flags & profiles.SQ_RNA) else api.SAMPLE_DNA)
    p_dev, p_host = str(tmp_path / "dev.blow5"), str(tmp_path / "host.blow5")
    w = api.Blow5Writer(p_dev, o.profile, o.flags, threads=2, stored=True)
    wh = api.Blow5Writer(p_host, o.profile, o.flags, threads=1, stored=True)
    done = 0
    while done < len(ids):
        nb = min(o.batch, len(ids) - done)
        b = gen.sample(nb).run().wait()
        w.write_batch(b, ids[done:done + nb])
        enc, eo = b.compress()
        wh.write(ids[done:done + nb], b.offset, b.median_before, b.sig_off, enc, eo)
        b.free()
        done += nb
    w.close(); wh.close()
    gen.close()
    got = open(p_dev, "rb").read()
    assert got == open(p_host, "rb").read()
    gold = os.path.join(GOLD, cid + ".blow5")
    assert parse_blow5(got) == parse_blow5(open(gold, "rb").read())
    _stored_records_are_valid(got)
    d = ref_dump(p_dev)
    assert d is None or d == ref_dump(gold)


@pytest.mark.gpu
def test_device_framing_of_a_bench_sized_batch(tmp_path):
    """2048 reads of 10 kb (records of 2-4 stored blocks): the device's records against the host framing, byte for byte, and read back
    by the reference's slow5lib (hashes of the signals against the batch's own)"""
    import bench
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    gen.load_genome(bench.synthetic_genome_host(8.0), 10000, api.SAMPLE_DNA)
    b = gen.sample(2048).run().wait()
    ids = [b"S1_%d!c!0!1!+" % (i + 1) for i in range(b.n_reads)]
    recs, ro = b.blow5_records(prof, fl, ids, read_number0=5, start_time0=12345)
    enc, eo = b.compress()
    ph = str(tmp_path / "h.blow5")
    wh = api.Blow5Writer(ph, prof, fl, threads=4, stored=True)
    wh.write(ids, b.offset, b.median_before, b.sig_off, enc, eo)
    wh.close()
    host = open(ph, "rb").read()
    hs = struct.unpack_from("<I", host, 64)[0]
    body = host[68 + hs:-5]
    assert len(recs) == ro[-1] == len(body)
    # (the host file starts its numbering at 0: patch nothing, compare a file written by write_batch instead)
    pd = str(tmp_path / "d.blow5")
    w = api.Blow5Writer(pd, prof, fl, threads=4, stored=True)
    w.write_batch(b, ids)
    n = w.close()
    dev = open(pd, "rb").read()
    assert n == len(dev) and dev == host
    _stored_records_are_valid(dev[:68 + hs] + dev[68 + hs:68 + hs + int(ro[3])] + b"5WOLB")       # (the first three records, parsed in Python)
    d = ref_dump(pd, "hash")
    if d is not None:
        sig = b.signal()
        lines = [ln for ln in d.splitlines() if ln.startswith("S1_")]
        assert len(lines) == b.n_reads
        for i in (0, 1, 777, b.n_reads - 1):
            h = 1469598103934665603
            for v in sig[b.sig_off[i]:b.sig_off[i + 1]].astype(np.uint16).tolist():
                h = ((h ^ (v & 0xff)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
                h = ((h ^ (v >> 8)) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            f = lines[i].split("\t")
            assert f[-1] == "fnv1a:%016x" % h and int(f[6]) == b.sig_off[i + 1] - b.sig_off[i] and int(f[9]) == i
    b.free(); gen.close()


# ---- SQG_BLOW5_SHARDS: the stored-block records on several files --------------------------------------------------------------------
def _records_of(paths):
    """header of the first file, and every file's uncompressed records, all files' in read_number order"""
    hdrs, recs = [], []
    for pth in paths:
        buf = open(pth, "rb").read()
        _stored_records_are_valid(buf)
        h, r = parse_blow5(buf)
        hdrs.append(h); recs += r

    def read_number(rec):
        (idl,) = struct.unpack_from("<H", rec, 0)
        q = 2 + idl + 4 + 32
        (nb,) = struct.unpack_from("<Q", rec, q)
        return struct.unpack_from("<i", rec, q + 8 + nb + 9 + 8)[0]
    assert all(h == hdrs[0] for h in hdrs)
    return hdrs[0], sorted(recs, key=read_number)


@pytest.mark.parametrize("cid", ["r9_t1", "r9_two_batches", "r9_ont"])
@pytest.mark.parametrize("shards", [2, 3])
def test_sharded_files_hold_the_reference_records(cid, shards, tmp_path):
    """SQG_BLOW5_SHARDS(n): n valid BLOW5 files whose records, put together in read_number order, are the reference file's; the
    reference's slow5lib reads every one of them"""
    o, ids, offset, median, so, sig = _case(cid)
    gold = os.path.join(GOLD, cid + ".blow5")
    encs = [orc.svb_zd(sig[so[i]:so[i + 1]]) for i in range(len(ids))]
    w = api.Blow5Writer(str(tmp_path / "s.blow5"), o.profile, o.flags, threads=2, stored=True, shards=shards)
    assert [os.path.basename(q) for q in w.paths] == ["s.%d.blow5" % i for i in range(shards)]
    done = 0
    while done < len(ids):
        nb = min(o.batch, len(ids) - done)
        e = encs[done:done + nb]
        eo = np.concatenate(([0], np.cumsum([len(x) for x in e]))).astype(np.int64)
        w.write(ids[done:done + nb], offset[done:done + nb], median[done:done + nb], so[done:done + nb + 1] - so[done], np.concatenate(e), eo)
        done += nb
    n = w.close()
    assert n == sum(os.path.getsize(q) for q in w.paths)
    hdr, recs = _records_of(w.paths)
    hdr_w, rec_w = parse_blow5(open(gold, "rb").read())
    assert hdr == hdr_w and recs == rec_w
    if os.path.exists(DUMP):
        lines = []
        for q in w.paths:
            d = ref_dump(q).splitlines()
            lines += [ln for ln in d if not ln.startswith(("@", "num_read_groups", "records"))]
        want = [ln for ln in ref_dump(gold).splitlines() if not ln.startswith(("@", "num_read_groups", "records"))]
        assert sorted(lines, key=lambda ln: int(ln.split("\t")[9])) == want
    with pytest.raises(api.SqgError):
        api.Blow5Writer(str(tmp_path / "z.blow5"), o.profile, o.flags, shards=2)           # several files: the stored-block mode only


@pytest.mark.gpu
def test_device_framed_records_on_four_files(tmp_path):
    """write_batch with SQG_BLOW5_SHARDS(4): the device's records dealt out to four files, written side by side behind the caller, three
    batches; together they are the one-file writer's records"""
    import bench
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    gen.load_genome(bench.synthetic_genome_host(8.0), 10000, api.SAMPLE_DNA)
    w4 = api.Blow5Writer(str(tmp_path / "q.blow5"), prof, fl, stored=True, shards=4)
    w1 = api.Blow5Writer(str(tmp_path / "one.blow5"), prof, fl, stored=True)
    nread = 0
    for n in (301, 64, 3):
        b = gen.sample(n).run().wait()
        ids = [b"S1_%d!c!0!1!+" % (nread + i + 1) for i in range(n)]
        w4.write_batch(b, ids); w1.write_batch(b, ids)
        nread += n
        b.free()
    n4, n1 = w4.close(), w1.close()
    gen.close()
    hdr4, rec4 = _records_of(w4.paths)
    hdr1, rec1 = _records_of(w1.paths)
    assert hdr4 == hdr1 and rec4 == rec1 and len(rec1) == nread
    assert n4 == n1 + 3 * (68 + len(hdr1) + 5)                      # (three more headers and end markers)
    d = [ref_dump(q, "hash") for q in w4.paths]
    assert d[0] is None or sum(int(x.strip().splitlines()[-1].split("\t")[1]) for x in d) == nread


# ---- the stored-mode writer and the context it reads from ----------------------------------------------------------------
@pytest.mark.gpu
def test_stored_writer_takes_a_batch_with_an_over_long_id_and_survives_its_context(tmp_path):
    """(1) a read id longer than the framing kernel's 4096-byte header does not fail the batch mid-file: that batch is framed on the host
    (sqg_blow5_write's stored path, ids up to 65535 bytes), the file holds the same records as an all-host stored file;
    (2) the context may be destroyed before the writer is closed -- the background write of the last batch reads the CONTEXT's pinned
    buffer: sqg_destroy drains and unbinds the writer first -- and the file is complete"""
    import bench
    prof, fl = profiles.get_profile("dna-r10-prom")
    mean, stdv = model.synthetic_model(9)
    gen = api.SignalGenerator(prof, fl, 9, mean, stdv, 42, num_workers=1, mode=api.MODE_CERTIFIED)
    gen.load_genome(bench.synthetic_genome_host(8.0), 4000, api.SAMPLE_DNA)
    pd, ph = str(tmp_path / "d.blow5"), str(tmp_path / "h.blow5")
    wd = api.Blow5Writer(pd, prof, fl, threads=2, stored=True)
    wh = api.Blow5Writer(ph, prof, fl, threads=2, stored=True)
    for bi in range(3):
        b = gen.sample(96).run().wait()
        ids = [b"S1_%d_%d!c!0!1!+" % (bi, i) for i in range(b.n_reads)]
        if bi == 1:
            ids[7] = b"L" * 5000                                  # > B5_ID_MAX: this batch goes through the host framing
        enc, eo = b.compress()
        wh.write(ids, b.offset, b.median_before, b.sig_off, enc, eo)
        wd.write_batch(b, ids)
        b.free()
    gen.close()                                                   # the context goes first: the last batch's records are still being written
    nd = wd.close()
    wh.close()
    dev, host = open(pd, "rb").read(), open(ph, "rb").read()
    assert nd == len(dev) and dev == host
    _stored_records_are_valid(dev)
    _, recs = parse_blow5(dev)
    assert len(recs) == 3 * 96 and struct.unpack_from("<H", recs[96 + 7], 0)[0] == 5000
