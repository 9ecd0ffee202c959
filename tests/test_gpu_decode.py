"""Parity tests proper: the HIP path, called through the C ABI (include/brotlig_amd.h), against the
CPU oracle on the same seeded inputs.  Bit-exact or fail.  Run on an MI355X with `-m gpu`."""
import hashlib
import json
import os

import numpy as np
import pytest

from brotli_g_sdk_amd import datagen as D
from brotli_g_sdk_amd import encoder as E
from cases import plain_cases, precon_cases, raw_stress_cases, symbol_overflow_cases
from helpers import oracle_decode

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a HIP device"
    from brotli_g_sdk_amd import api as a
    a.lib()                        # fails loudly if libbrotlig_hip.so is missing
    return a


_COMPANION = []


def _companion():
    """A stream of another kind (text, two pages) that shares the wavefront with the case under test in the two-pages-per-wavefront form."""
    if not _COMPANION:
        d = D.text(65536 + 4321, 99)
        _COMPANION.append((d, E.encode(d)))
    return _COMPANION[0]


def decode_in_form(api, form, stream, out_size=None):
    """The decoded bytes of `stream` under the kernel form the `kernel_form` fixture has pinned (tests/conftest.py).  `rule` and `solo` go
    through the host-pointer entry (DecodeGPU), like a caller of the reference; `pair` -- two pages per wavefront, the form of every large
    batch -- needs more than one page in the launch: the stream twice, a companion of another kind between them, one wavefront for all."""
    if form not in ("pair", "pair3"):
        out, ms = api.DecodeGPU(stream, output_size=out_size)
        assert ms > 0.0
        return out
    comp_d, comp_s = _companion()
    n = int(api.DecompressedSize(stream)) if out_size is None else int(out_size)
    # pair3: three wavefronts, at least seven pages (the companion has two) -- all three run the two-page form and hand pages out between them
    copies = 3 if form == "pair3" else 2
    streams, sizes = [], []
    for k in range(copies):
        streams += [stream] + ([comp_s] if k + 1 < copies else [])
        sizes += [n] + ([len(comp_d)] if k + 1 < copies else [])
    dec = api.BatchDecoder(streams, out_sizes=sizes)
    dec.poison_output()
    dec.decode()
    for k in range(copies - 1):
        assert np.array_equal(dec.output(2 * k + 1), comp_d), "the companion stream"
    first = dec.output(0)
    for k in range(1, copies):
        assert np.array_equal(first, dec.output(2 * k)), "the same stream several times in one launch"
    return first


def test_wave_primitives_on_device(api):
    """DPP half-wave scan vs shuffle scan, half ballot / shuffle / max, checked on the host."""
    api.DeviceSelfTest()


@pytest.mark.parametrize("name,thunk,kw", plain_cases(), ids=[c[0] for c in plain_cases()])
def test_decode_gpu_plain(api, kernel_form, name, thunk, kw):
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    out = decode_in_form(api, kernel_form, stream)
    assert len(out) == len(ref) and np.array_equal(out, ref)


@pytest.mark.parametrize("name,thunk,pre", precon_cases(), ids=[c[0] for c in precon_cases()])
def test_decode_gpu_preconditioned(api, kernel_form, name, thunk, pre):
    tex = thunk()
    stream = E.encode(tex, precondition=pre)
    rc, ref = oracle_decode(stream, out_size=len(tex))
    assert rc == 0 and np.array_equal(ref, tex)
    out = decode_in_form(api, kernel_form, stream, out_size=len(tex))
    assert np.array_equal(out, ref)


@pytest.mark.parametrize("name,thunk,kw", raw_stress_cases(), ids=[c[0] for c in raw_stress_cases()])
def test_far_copies_read_what_the_previous_group_flushed(api, name, thunk, kw):
    """Global read-after-write inside a wavefront: the decoder keeps 656 bytes of history on chip (528 until round 3) and flushes
    the window group by group; copies from 513..1400 bytes back read global memory that the same wavefront stored one
    group earlier, with no fence in between (the design argument is in brotlig_kernels.h, step 3b).  Four streams
    side by side so that several wavefronts run the pattern at once; bit-exact against the oracle."""
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    dec = api.BatchDecoder([stream] * 4)
    # a batch this small is decoded two wavefronts per page (brotlig_decode_duo_kernel); the second pass takes the one-wavefront kernel
    # with one page per wavefront (its own instantiation of the page loop, with its own group / window geometry), the third forces
    # three wavefronts, so that the same pages also go two to a wavefront like those of a large batch
    for mode, grid in ((0, 0), (1, 0), (1, 3)):
        api.DebugSetDecodeMode(mode)
        api.DebugSetDecodeGrid(grid)
        try:
            for _ in range(3 if grid == 0 else 2):
                dec.poison_output()
                dec.decode()
                for i in range(4):
                    assert np.array_equal(dec.output(i), ref), (name, i, mode, grid)
        finally:
            api.DebugSetDecodeGrid(0)
            api.DebugSetDecodeMode(0)


@pytest.mark.parametrize("name,thunk,kw", symbol_overflow_cases(), ids=[c[0] for c in symbol_overflow_cases()])
def test_prefix_codes_with_more_symbols_than_the_lds_arrays_hold(api, kernel_form, name, thunk, kw):
    """~300 distinct ICP or distance symbols in a page: the decoder keeps the first 255 / 96 (canonical order) in LDS
    and reads the rest from its workspace.  Several copies side by side, decoded twice: the global slots are reused
    page after page and must never serve a previous page's symbols."""
    data = thunk()
    stream = E.encode(data, **kw)
    rc, ref = oracle_decode(stream)
    assert rc == 0 and np.array_equal(ref, data)
    other = E.encode(D.text(3 * 65536, 77))
    dec = api.BatchDecoder([stream, other, stream, other, stream])
    for _ in range(2):
        dec.poison_output()
        dec.decode()
        for i in (0, 2, 4):
            assert np.array_equal(dec.output(i), ref), (name, i)


def test_golden_fixtures_on_gpu(api, kernel_form):
    index = json.load(open(os.path.join(GOLDEN, "index.json")))
    names = sorted(index)
    streams = [np.fromfile(os.path.join(GOLDEN, name + ".brotlig"), dtype=np.uint8) for name in names]
    if kernel_form in ("pair", "pair3"):       # all fixtures in ONE launch on one (three) wavefront(s): pages of unrelated fixtures side by side in the halves
        dec = api.BatchDecoder(streams, out_sizes=[index[n]["size"] for n in names])
        dec.poison_output()
        dec.decode()
        outs = [dec.output(i) for i in range(len(names))]
    else:
        outs = [api.DecodeGPU(s, output_size=index[n]["size"])[0] for n, s in zip(names, streams)]
    for name, out in zip(names, outs):
        assert hashlib.sha256(out.tobytes()).hexdigest() == index[name]["sha256"], (name, kernel_form)


def test_error_codes(api):
    """Same two checks as src/BrotligDecoder.cpp:437-446."""
    stream = E.encode(np.zeros(1000, np.uint8))
    bad = stream.copy(); bad[1] ^= 0x10
    with pytest.raises(api.BrotligError) as e:
        api.DecodeGPU(bad, output_size=1000)
    assert e.value.code == api.BROTLIG_ERROR_CORRUPT_STREAM
    bad = stream.copy(); bad[0] = 6; bad[1] = 6 ^ 0xFF
    with pytest.raises(api.BrotligError) as e:
        api.DecodeGPU(bad, output_size=1000)
    assert e.value.code == api.BROTLIG_ERROR_INCORRECT_STREAM_FORMAT


def test_batch_of_streams(api):
    """One launch, many streams of unequal page counts (pages of different streams share a wave)."""
    datas = [D.text(65536 + 100, 1), D.runs(3 * 65536, 2), D.random_bytes(65536, 3), D.records(5 * 65536 + 1, 4),
             D.mixed(65536, 5), D.samples16(9 * 65536, 6), np.zeros(1, np.uint8)]
    streams = [E.encode(d) for d in datas]
    dec = api.BatchDecoder(streams)
    dec.poison_output()
    dec.decode()
    for i, d in enumerate(datas):
        assert np.array_equal(dec.output(i), d), i


def test_batch_of_twenty_thousand_small_streams(api):
    """What an asset streamer hands over: thousands of small streams in one batch.  20 011 streams of 1 byte .. a page and a bit (96 distinct
    ones, cycled), two of them damaged: the prepare kernel walks every header (32 per step), the page-to-stream search runs over 20 011
    entries, the per-stream status comes back as one strided copy of 20 011 words -- every stream bit-exact, the two damaged ones named."""
    rng = np.random.default_rng(91)
    makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
    base = [makers[i % 6](int(rng.integers(1, 65536 + 3000)) if i % 5 else int(rng.integers(1, 40)), 3000 + i) for i in range(96)]
    enc = [E.encode(d) for d in base]
    n = 20011
    pick = rng.integers(0, 96, n)
    streams = [enc[k] for k in pick]
    bad_a, bad_b = 7, n - 2
    hb = streams[bad_a].copy(); hb[0] ^= 0x10; streams[bad_a] = hb                              # stream id damaged: refused by the prepare kernel
    big = next(k for k in range(96) if len(base[k]) > 4096 and k % 6 != 5)                       # a compressed page, not a stored one
    pb = enc[big].copy(); pick[bad_b] = big
    pb[8 + 4 + 40:8 + 4 + 60] ^= 0x5A                                                            # bytes inside the first page (any result, no fault)
    streams[bad_b] = pb
    dec = api.BatchDecoder(streams, out_sizes=[len(base[k]) for k in pick])
    dec.poison_output()
    try:
        dec.decode()
    except api.BrotligError:
        pass
    rc, per = dec.stream_status()
    assert len(per) == n and per[bad_a] == api.BROTLIG_ERROR_CORRUPT_STREAM
    assert [i for i, r in enumerate(per) if r != api.BROTLIG_OK and i not in (bad_a, bad_b)] == []
    out = dec.d_out.cpu().numpy()
    for i in range(n):
        if i not in (bad_a, bad_b):
            d = base[pick[i]]
            assert np.array_equal(out[dec.out_offs[i]:dec.out_offs[i] + len(d)], d), i


def test_batch_mixed_plain_and_preconditioned(api):
    tex = D.bc_texture(3, 96, 64, seed=5)
    pre = dict(format=3, width_blocks=96, height_blocks=64, swizzle=1, delta=1)
    datas = [D.text(2 * 65536, 1), tex, D.runs(65536 + 5, 2)]
    streams = [E.encode(datas[0]), E.encode(tex, precondition=pre), E.encode(datas[2])]
    dec = api.BatchDecoder(streams, out_sizes=[len(d) for d in datas])
    dec.poison_output()
    dec.decode()
    for i, d in enumerate(datas):
        rc, ref = oracle_decode(streams[i], out_size=len(d))
        assert rc == 0 and np.array_equal(dec.output(i), ref), i


def test_kernel_choice_follows_the_page_count(api):
    """Up to 2 048 pages a batch is decoded two wavefronts per page (brotlig_decode_duo_kernel), beyond that one wavefront per one or
    two pages (brotlig_decode_kernel).  The host only knows the output size: for 64 .. 256 MiB both kernels are launched and each
    reads the page count the prepare kernel found (DecodeArgs::duo_limit) -- exactly one of them decodes.  Page counts on both sides
    of the limit and of the sizes at which the host stops launching one of the two; every tiled repeat against the source bytes,
    under the size rule and with either kernel pinned."""
    base = D.mixed(32 * 65536, 21)
    small = E.encode(base)
    for reps in (1, 9, 33, 63, 64, 65, 100, 129):        # 32 .. 4 128 pages, 2 .. 258 MiB
        stream = D.tile_stream(small, reps)
        dec = api.BatchDecoder([stream])
        for mode in (0, 1, 2) if reps in (1, 65) else (0,):
            api.DebugSetDecodeMode(mode)
            try:
                dec.poison_output()
                dec.decode()
                out = dec.output(0)
                assert np.array_equal(out.reshape(reps, -1), np.broadcast_to(base, (reps, len(base)))), (reps, mode)
            finally:
                api.DebugSetDecodeMode(0)
        del dec


def test_repeated_decode_is_idempotent(api):
    d = D.mixed(8 * 65536, 3)
    dec = api.BatchDecoder([E.encode(d)])
    for _ in range(3):
        dec.poison_output()
        dec.decode()
        assert np.array_equal(dec.output(0), d)


def test_config2_full_size_runs(api):
    """BASELINE config 2: 256 MiB = 4096 pages of zeros + byte runs, one stream, bit-exact vs the oracle."""
    d = D.runs(256 * 65536, 1)
    small = E.encode(d)
    stream = D.tile_stream(small, 16)
    assert (int(stream[2]) | (int(stream[3]) << 8)) == 4096
    rc, ref = oracle_decode(stream)
    assert rc == 0 and len(ref) == 256 * 2**20
    dec = api.BatchDecoder([stream])
    dec.poison_output()
    dec.decode()
    out = dec.output(0)
    assert np.array_equal(out, ref)
    assert np.array_equal(out.reshape(16, -1), np.broadcast_to(d, (16, len(d))))


def test_config3_shape_mixed_many_streams(api):
    """Config-3 shaped batch at reduced size (8 streams x 512 pages): checksum of checksums vs the oracle,
    plus page-level periodicity (tiled pages decode to identical bytes)."""
    streams, refs = [], []
    for k in range(8):
        d = D.mixed(64 * 65536, 50 + k)
        s = D.tile_stream(E.encode(d), 8)
        rc, ref = oracle_decode(s)
        assert rc == 0
        streams.append(s); refs.append(ref)
    dec = api.BatchDecoder(streams)
    dec.poison_output()
    dec.decode()
    agg_gpu, agg_ref = hashlib.sha256(), hashlib.sha256()
    for k in range(8):
        out = dec.output(k)
        agg_gpu.update(hashlib.sha256(out.tobytes()).digest())
        agg_ref.update(hashlib.sha256(refs[k].tobytes()).digest())
        assert np.array_equal(out.reshape(8, -1)[0], out.reshape(8, -1)[7])
    assert agg_gpu.hexdigest() == agg_ref.hexdigest()


def test_config4_bc3_texture_stream(api):
    """Config-4 shaped input at reduced size: a BC3 texture of 256 x 256 blocks (1 MiB, 16 pages),
    swizzle + delta on ("BC7-style" is realised as BC3: the reference has BC1-BC5 only)."""
    tex = D.bc_texture(3, 256, 256, seed=9)
    pre = dict(format=3, width_blocks=256, height_blocks=256, swizzle=1, delta=1)
    s = E.encode(tex, precondition=pre)
    rc, ref = oracle_decode(s, out_size=len(tex))
    assert rc == 0 and np.array_equal(ref, tex)
    dec = api.BatchDecoder([s, s, s], out_sizes=[len(tex)] * 3)
    dec.poison_output()
    dec.decode()
    for i in range(3):
        assert np.array_equal(dec.output(i), ref)


@pytest.mark.gpu
def test_streamer_overlapped_batches(api):
    """Streaming front end (include/brotlig_amd.h BrotligStreamer*): batches of mixed plain and
    pre-conditioned streams go through a 3-slot ring, more batches than slots, results read back both
    ways (copied to caller buffers, and from the pinned slot); every byte is checked against the oracle."""
    batches = []
    for b in range(7):
        datas = [D.mixed(65536 * 3 + 1000 * b, 100 + b), D.text(40000 + b, 200 + b), D.runs(65536 * 2, 300 + b)]
        streams = [E.encode(d) for d in datas]
        tex = D.bc_texture(1 + b % 5, 64, 32, seed=b)
        streams.append(E.encode(tex, precondition=dict(format=1 + b % 5, width_blocks=64, height_blocks=32, swizzle=True, delta=True)))
        batches.append(streams)
    st = api.Streamer(slots=3, slot_in_bytes=4 << 20, slot_out_bytes=8 << 20, max_streams=16)
    tickets, user = [], []
    for k, streams in enumerate(batches):
        outs = None
        if k % 2 == 0:
            outs = [np.full(api.DecompressedSize(s), 0xAB, np.uint8) for s in streams]
        tickets.append(st.submit(streams, outs))
        user.append(outs)
        if k >= 2:                                  # stay two batches ahead of the reader
            j = k - 2
            _check_batch(st, tickets[j], batches[j], user[j])
    for j in range(len(batches) - 2, len(batches)):
        _check_batch(st, tickets[j], batches[j], user[j])
    with pytest.raises(api.BrotligError):           # a batch larger than a slot is refused, not truncated
        st.submit([E.encode(D.runs(65536 * 200, 1))])
    bad = batches[0][0].copy(); bad[1] ^= 0xFF
    with pytest.raises(api.BrotligError):
        st.submit([bad])
    st.close()


def _check_batch(st, ticket, streams, user_outs):
    if user_outs is not None:
        st.wait(ticket)
        got = user_outs
    else:
        got = st.result(ticket)
    for s, g in zip(streams, got):
        rc, ref = oracle_decode(s)
        assert rc == 0 and np.array_equal(g, ref)


@pytest.mark.gpu
def test_page_schedule_large_batch(api):
    """Batches of 768 MiB and more are decoded through the page schedule (pages grouped into size buckets,
    dense first: include/brotlig_amd.h BrotligDecodeWorkspaceSizeFor).  1 GiB of four different page kinds
    plus a pre-conditioned texture: the output is the same with the schedule and without it (minimum
    workspace), and equals the bytes the streams were made from."""
    kinds = [D.mixed(128 * 65536, 21), D.runs(128 * 65536, 22), D.text(128 * 65536, 23), D.records(128 * 65536, 24)]
    streams = [D.tile_stream(E.encode(d), 32) for d in kinds]           # 4 x 4096 pages = 1 GiB
    tex = D.bc_texture(3, 128, 128, seed=25)
    streams.append(E.encode(tex, precondition=dict(format=3, width_blocks=128, height_blocks=128, swizzle=True, delta=True)))
    sizes = [32 * len(d) for d in kinds] + [len(tex)]
    outs = []
    for schedule in (True, False):
        dec = api.BatchDecoder(streams, out_sizes=sizes, schedule=schedule)
        dec.poison_output()
        dec.decode()
        outs.append([dec.output(i) for i in range(len(streams))])
        del dec
    for i, d in enumerate(kinds):
        assert np.array_equal(outs[0][i].reshape(32, -1), np.broadcast_to(d, (32, len(d))))
        assert np.array_equal(outs[0][i], outs[1][i])
    assert np.array_equal(outs[0][4], tex) and np.array_equal(outs[1][4], tex)


@pytest.mark.gpu
def test_streamer_device_output_mode(api):
    """Round 6: a streamer whose decoded bytes STAY in device memory -- no download, no pinned staging for them.  Seven
    batches of mixed plain and pre-conditioned streams through a 2-slot ring (more batches than slots), one of them with a damaged stream: every
    stream is handed back as {device pointer, size, event} straight after Submit, a consumer on its own torch stream waits for the event and
    copies the bytes aside (its hand-back, BrotligStreamerConsumerDone, is what the slot's next batch waits for on the device); afterwards
    every byte is compared with the oracle and the damaged stream is named."""
    import torch
    from fuzzcases import simple_code_one_symbol
    bad, cap = simple_code_one_symbol()
    batches, wants = [], []
    for b in range(7):
        datas = [D.mixed(65536 * 3 + 1000 * b, 100 + b), D.text(40000 + b, 200 + b), D.runs(65536 * 2, 300 + b)]
        streams = [E.encode(d) for d in datas]
        tex = D.bc_texture(1 + b % 5, 64, 32, seed=b)
        streams.append(E.encode(tex, precondition=dict(format=1 + b % 5, width_blocks=64, height_blocks=32, swizzle=True, delta=True)))
        want = []
        for s_, d in zip(streams, datas + [tex]):
            rc, ref = oracle_decode(s_, out_size=len(d))
            assert rc == 0
            want.append(ref)
        if b == 4:
            streams.insert(1, bad); want.insert(1, None)
        batches.append(streams); wants.append(want)
    st = api.Streamer(slots=2, slot_in_bytes=4 << 20, slot_out_bytes=8 << 20, max_streams=16, device_output=True)
    with pytest.raises(api.BrotligError):               # nothing is downloaded in this mode: caller buffers are refused
        st.submit(batches[0], [np.zeros(api.DecompressedSize(s_), np.uint8) for s_ in batches[0]])
    consumer = torch.cuda.Stream()
    kept, tickets = [], []
    for k, streams in enumerate(batches):
        t = st.submit(streams)                          # (with both slots in flight this completes the oldest batch first)
        tickets.append(t)
        with torch.cuda.stream(consumer):
            copies = []
            for i in range(len(streams)):
                ten = st.device_tensor(t, i)            # makes `consumer` wait for the batch's event
                copies.append(ten.clone())
            st.consumer_done(t)
        kept.append(copies)
        assert st.output(t, 0) is None                  # no host copy exists
    for k, (t, want) in enumerate(zip(tickets, wants)):
        if k >= len(tickets) - 2:                       # the last two batches are still in their slots
            if k == 4:
                with pytest.raises(api.BrotligError):
                    st.wait(t)
            else:
                st.wait(t)
    assert st.stream_results(tickets[-1], len(batches[-1])) == [api.BROTLIG_OK] * len(batches[-1])
    consumer.synchronize()
    for k, (copies, want) in enumerate(zip(kept, wants)):
        for i, (c, w) in enumerate(zip(copies, want)):
            if w is not None:
                assert np.array_equal(c.cpu().numpy(), w), (k, i)
    st.close()
    # the damaged stream is named while its batch is still in its slot
    st = api.Streamer(slots=2, slot_in_bytes=4 << 20, slot_out_bytes=8 << 20, max_streams=16, device_output=True)
    t = st.submit(batches[4])
    with pytest.raises(api.BrotligError):
        st.wait(t)
    res = st.stream_results(t, len(batches[4]))
    assert res[1] == api.BROTLIG_ERROR_GENERIC and all(r == api.BROTLIG_OK for i, r in enumerate(res) if i != 1)
    for i, w in enumerate(wants[4]):
        if w is not None:
            assert np.array_equal(st.device_tensor(t, i).cpu().numpy(), w), i
    st.close()


@pytest.mark.parametrize("device_output", [False, True], ids=["host_output", "device_output"])
def test_streamer_streams_written_in_place(api, device_output):
    """Round 6: BrotligStreamerAcquire / SubmitInPlace -- the compressed streams are written straight into the slot's pinned staging area (what a
    file read does) instead of being copied there by Submit.  Five batches through two slots, both output modes; a batch with a bad offset and
    one with a damaged header are refused and leave the area acquired; an ordinary Submit is refused while an area is out."""
    st = api.Streamer(slots=2, slot_in_bytes=2 << 20, slot_out_bytes=4 << 20, max_streams=8, device_output=device_output)
    tickets, wants = [], []
    for b in range(5):
        datas = [D.mixed(65536 + 999 * b, 40 + b), D.text(30000 + b, 50 + b), D.runs(70000, 60 + b)]
        streams = [E.encode(d) for d in datas]
        area = st.acquire()
        offs, pos = [], 0
        for s_ in streams:
            offs.append(pos); area[pos:pos + len(s_)] = s_; pos = (pos + len(s_) + 15) // 16 * 16
        if b == 1:
            with pytest.raises(api.BrotligError):
                st.submit(streams)                                  # an area is out: the ordinary Submit waits its turn
            with pytest.raises(api.BrotligError):
                st.submit_in_place([offs[0] + 4] + offs[1:], [len(s_) for s_ in streams])      # not 16-byte aligned
            keep = area[1]; area[1] ^= 0x10
            with pytest.raises(api.BrotligError):
                st.submit_in_place(offs, [len(s_) for s_ in streams])                           # damaged magic: refused, still acquired
            area[1] = keep
        outs = None if (device_output or b % 2) else [np.full(len(d), 0xAB, np.uint8) for d in datas]
        tickets.append((st.submit_in_place(offs, [len(s_) for s_ in streams], outs), outs)); wants.append(datas)
        if b >= 1:
            t, o = tickets[b - 1]
            if device_output:
                got = [st.device_tensor(t, i).cpu().numpy() for i in range(3)]
                st.consumer_done(t)
                st.wait(t)
            else:
                got = st.result(t)
            for g, w in zip(got, wants[b - 1]):
                assert np.array_equal(g, w), b
    t, o = tickets[-1]
    got = [st.device_tensor(t, i).cpu().numpy() for i in range(3)] if device_output else st.result(t)
    for g, w in zip(got, wants[-1]):
        assert np.array_equal(g, w)
    st.close()


def test_schedule_kernel_in_garbage_and_reused_workspaces(api):
    """Round 6: the one kernel in front of the page decode needs no memset -- it initialises a workspace it has never seen (garbage where its
    cookie, tickets and counters live: one workgroup wins the right to zero them, the others wait for it) and leaves its words clean for the
    next launch.  A batch on the ticket path (80 streams: prepare, scan, finalize, count, scatter items) and one on the one-workgroup path, each
    in a workspace of random bytes, then again in the same workspace, then in a workspace that holds what ANOTHER batch of another shape left
    behind (header words copied over); every stream bit-exact, the damaged one named, every time."""
    import torch
    from fuzzcases import simple_code_one_symbol
    bad, cap = simple_code_one_symbol()
    rng = np.random.default_rng(61)
    makers = [D.text, D.records, D.samples16, D.runs, D.mixed, D.random_bytes]
    big_d = [makers[i % 6](int(rng.integers(1, 3 * 65536)), 7000 + i) for i in range(80)]
    big_s = [E.encode(d) for d in big_d]
    big_s[17] = bad
    sizes = [len(d) for d in big_d]; sizes[17] = cap
    small_d = [D.mixed(5 * 65536 + 77, 1), D.text(65536, 2), D.runs(2 * 65536, 3)]
    small_s = [E.encode(d) for d in small_d]
    decs = {"big": api.BatchDecoder(big_s, out_sizes=sizes), "small": api.BatchDecoder(small_s)}

    def check(name):
        dec = decs[name]
        dec.poison_output()
        if name == "big":
            with pytest.raises(api.BrotligError):
                dec.decode()
            rc, per = dec.stream_status()
            assert [i for i, r in enumerate(per) if r != api.BROTLIG_OK] == [17]
            for i, d in enumerate(big_d):
                if i != 17:
                    assert np.array_equal(dec.output(i), d), i
        else:
            dec.decode()
            for i, d in enumerate(small_d):
                assert np.array_equal(dec.output(i), d), i

    g = torch.Generator(device="cuda"); g.manual_seed(5)
    for name in ("big", "small"):
        dec = decs[name]
        dec.d_ws.copy_(torch.randint(0, 256, (dec.ws_bytes,), dtype=torch.uint8, device="cuda", generator=g))     # a workspace nobody has zeroed
        check(name)
        check(name)                                     # the same workspace again: what the first launch left
    # what another batch of another shape left behind: the header words (status, counters, the schedule kernel's own) of the other workspace
    hdr = 192 * 4
    a, b = decs["big"].d_ws[:hdr].clone(), decs["small"].d_ws[:hdr].clone()
    decs["big"].d_ws[:hdr].copy_(b); decs["small"].d_ws[:hdr].copy_(a)
    check("big"); check("small"); check("big")


def test_streamer_ring_overflow_keeps_results(api):
    """More batches submitted than there are slots before anything is waited for: Submit completes the oldest
    batch to make room, fills its outputs[] and keeps its result for a later Wait.  A refused batch
    (bad header) must not disturb the batch whose slot it would have taken."""
    slots = 2
    st = api.Streamer(slots=slots, slot_in_bytes=2 << 20, slot_out_bytes=4 << 20, max_streams=8)
    datas = [[D.mixed(65536 + 777 * b, 500 + b), D.text(30000 + b, 600 + b)] for b in range(5)]
    streams = [[E.encode(d) for d in batch] for batch in datas]
    outs = [[np.full(len(d), 0xAB, np.uint8) for d in batch] for batch in datas]
    tickets = [st.submit(streams[b], outs[b]) for b in range(3)]            # batch 0 is displaced by batch 2
    bad = streams[0][0].copy(); bad[1] ^= 0xFF
    with pytest.raises(api.BrotligError):
        st.submit([bad])                                                    # refused: the ring stays as it was
    st.wait(tickets[0])                                                     # displaced batch: result still available
    for d, o in zip(datas[0], outs[0]):
        assert np.array_equal(o, d)
    tickets.append(st.submit(streams[3], outs[3]))                          # displaces batch 1
    tickets.append(st.submit(streams[4], None))                             # displaces batch 2; bytes stay pinned
    for b in (1, 2, 3):
        st.wait(tickets[b])
        for d, o in zip(datas[b], outs[b]):
            assert np.array_equal(o, d), b
    got = st.result(tickets[4])
    for d, o in zip(datas[4], got):
        assert np.array_equal(o, d)
    with pytest.raises(api.BrotligError):                                   # two generations back: forgotten
        st.wait(tickets[0])
    st.close()


def test_multi_device_entry_with_three_shards_on_the_one_device(api):
    """BrotligDecodeBatchMultiDevice (include/brotlig_amd.h): the stream list cut by BrotligShardPlan (balanced by
    compressed bytes), one BrotligDeviceBatch per shard, one host thread per shard.  The box has one device, so the
    shards share it on streams of their own; a node with G devices runs the same call with device = 0..G-1."""
    import torch
    datas = [D.mixed(3 * 65536 + 11 * k, 500 + k) for k in range(5)] + [D.runs(6 * 65536, 9), D.random_bytes(2 * 65536, 3),
                                                                        D.text(65536 + 77, 8), D.records(4 * 65536, 2)]
    streams = [E.encode(d) for d in datas]
    first = api.ShardPlan([len(s) for s in streams], 3)
    assert first[0] == 0 and first[-1] == len(streams) and all(b > a for a, b in zip(first, first[1:]))
    decs = [api.BatchDecoder(streams[a:b]) for a, b in zip(first, first[1:])]
    side = [torch.cuda.Stream() for _ in decs]
    for d in decs:
        d.poison_output()
    torch.cuda.synchronize()
    mk, mw, per = api.DecodeBatchMultiDevice(decs, warmup=1, steps=2, streams=[s.cuda_stream for s in side])
    assert all(r == 0 for r, _, _ in per) and mk == max(k for _, k, _ in per) > 0.0 and mw >= mk * 2 * 0.5
    torch.cuda.synchronize()
    k = 0
    for d in decs:
        for i in range(d.n):
            assert np.array_equal(d.output(i), datas[k]), k
            k += 1
    assert k == len(datas)
    # a damaged stream in one shard is that shard's result, and the call's
    bad = streams[0].copy(); bad[1] ^= 1
    decs2 = [api.BatchDecoder([bad]), api.BatchDecoder([streams[1]])]
    with pytest.raises(api.BrotligError):
        api.DecodeBatchMultiDevice(decs2)
    torch.cuda.synchronize()
    assert np.array_equal(decs2[1].output(0), datas[1])
    # the non-blocking pair: enqueue all shards, do something else, collect the status
    for d in decs:
        d.poison_output()
    torch.cuda.synchronize()
    job = api.MultiDeviceAsync(decs, streams=[s.cuda_stream for s in side])
    filler = torch.ones(1 << 20, device="cuda").sum()                      # the caller's own work, meanwhile
    assert job.wait() == [0, 0, 0] and float(filler) == float(1 << 20)
    k = 0
    for d in decs:
        for i in range(d.n):
            assert np.array_equal(d.output(i), datas[k]), k
            k += 1
    job2 = api.MultiDeviceAsync(decs2)
    with pytest.raises(api.BrotligError):
        job2.wait()


def test_context_decodes_assets_of_growing_and_shrinking_size(api):
    """BrotligContext: DecodeGPU with buffers kept between calls (they only grow)."""
    ctx = api.Context()
    for n, seed in ((70000, 1), (9 * 65536 + 5, 2), (100, 3), (3 * 65536, 4)):
        data = D.mixed(n, seed)
        out, ms = ctx.DecodeGPU(E.encode(data))
        assert np.array_equal(out, data) and ms > 0.0
    tex = D.bc_texture(1, 40, 24, seed=5)
    s = E.encode(tex, precondition=dict(format=1, width_blocks=40, height_blocks=24, swizzle=True, delta=True))
    out, _ = ctx.DecodeGPU(s, output_size=len(tex))
    rc, ref = oracle_decode(s, out_size=len(tex))
    assert rc == 0 and np.array_equal(out, ref)
    bad = E.encode(D.text(70000, 1)); bad[1] ^= 1
    with pytest.raises(api.BrotligError):
        ctx.DecodeGPU(bad)
    out, _ = ctx.DecodeGPU(E.encode(D.text(70000, 1)))              # still usable after an error
    assert np.array_equal(out, D.text(70000, 1))
    ctx.close()


def test_simple_code_with_one_symbol_rejects_the_page_on_device(api):
    """Same case as tests/test_sim_decode.py / test_cpu_decode.py: undefined in the format, rejected everywhere."""
    from fuzzcases import simple_code_one_symbol
    bad, cap = simple_code_one_symbol()
    good = D.text(70000, 4)
    dec = api.BatchDecoder([bad, E.encode(good)], out_sizes=[cap, len(good)])
    dec.poison_output()
    with pytest.raises(api.BrotligError):
        dec.decode()
    assert np.all(dec.output(0) == 0xCD) and np.array_equal(dec.output(1), good)


def test_per_stream_status_names_the_damaged_streams(api, kernel_form):
    """BrotligDecodeBatchStreamStatus (round 5): a 64-stream batch with three streams damaged in three ways -- a page refused when it is
    fetched, a page refused at its end, a stream refused by the prepare kernel -- names exactly those three, with the reference's error codes,
    in every form of the page kernel; the other 61 streams are bit-exact."""
    from fuzzcases import damaged_batch_for_stream_status
    streams, sizes, datas, expect = damaged_batch_for_stream_status()
    dec = api.BatchDecoder(streams, out_sizes=sizes)
    dec.poison_output()
    with pytest.raises(api.BrotligError):
        dec.decode()
    rc, per = dec.stream_status()
    assert rc == api.BROTLIG_ERROR_CORRUPT_STREAM                    # the batch-wide answer: the header failure wins, as before
    want = {i: (api.BROTLIG_ERROR_CORRUPT_STREAM if bit == 1 else api.BROTLIG_ERROR_GENERIC) for i, bit in expect.items()}
    assert {i: r for i, r in enumerate(per) if r != api.BROTLIG_OK} == want
    for i, d in enumerate(datas):
        if d is not None:
            assert np.array_equal(dec.output(i), d), i
    # a clean batch afterwards in the same workspace: every word is back to OK
    good = [s for s, d in zip(streams, datas) if d is not None][:5]
    dec2 = api.BatchDecoder(good)
    dec2.decode()
    assert dec2.stream_status() == (api.BROTLIG_OK, [api.BROTLIG_OK] * 5)


def test_streamer_reports_the_damaged_stream_and_delivers_the_others(api):
    """BrotligStreamerStreamResult (round 5): Wait still fails for a batch with a damaged stream, but the batch's other streams are
    delivered (outputs[] filled, BrotligStreamerOutput serves them) and the damaged one is named."""
    from fuzzcases import simple_code_one_symbol
    bad, cap = simple_code_one_symbol()
    datas = [D.text(70000, 1), None, D.mixed(3 * 65536 + 5, 2), D.runs(65536, 3)]
    streams = [E.encode(d) if d is not None else bad for d in datas]
    st = api.Streamer(slots=2, slot_in_bytes=2 << 20, slot_out_bytes=4 << 20, max_streams=8)
    outs = [np.full(len(d) if d is not None else cap, 0xAB, np.uint8) for d in datas]
    t = st.submit(streams, outs)
    with pytest.raises(api.BrotligError):
        st.wait(t)
    assert st.stream_results(t, 4) == [api.BROTLIG_OK, api.BROTLIG_ERROR_GENERIC, api.BROTLIG_OK, api.BROTLIG_OK]
    for d, o in zip(datas, outs):
        if d is not None:
            assert np.array_equal(o, d)
    assert np.all(outs[1] == 0xAB)                                   # nothing of the damaged stream is handed over
    assert st.output(t, 1) is None and np.array_equal(st.output(t, 2), datas[2])
    t2 = st.submit([streams[0]])                                     # the streamer goes on
    assert np.array_equal(st.result(t2)[0], datas[0])
    st.close()

