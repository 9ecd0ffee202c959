"""On-device differential and negative tests: seeded random streams -- random data
compositions, encoder flags, page sizes, NPOSTFIX / NDIRECT, pre-conditioning parameters -- through the
HIP path (C ABI, BrotligDecodeBatchDevice / DecodeGPU) against the CPU oracle, bit-exact; and damaged
streams, which must come back as a status (the reference only checks the two header bytes,
src/BrotligDecoder.cpp:437-446, and is undefined beyond that) without faulting the device, leaving the
next valid decode bit-exact."""
import numpy as np
import pytest

from brotli_g_sdk_amd import encoder as E
from fuzzcases import corrupt, random_plain, random_precon
from helpers import oracle_decode

pytestmark = pytest.mark.gpu

N_PLAIN = 220
N_PRECON = 60


@pytest.fixture(scope="module")
def api():
    import torch
    assert torch.cuda.is_available(), "the gpu tests need a HIP device"
    from brotli_g_sdk_amd import api as a
    a.lib()
    return a


@pytest.mark.parametrize("chunk", range(0, N_PLAIN, 44))
def test_random_streams_batched(api, kernel_form, chunk):
    """44 random streams per launch (pages of unrelated streams share wavefronts), each compared with the oracle."""
    datas, streams = [], []
    for seed in range(chunk, chunk + 44):
        d, kw = random_plain(seed)
        datas.append(d)
        streams.append(E.encode(d, **kw))
    dec = api.BatchDecoder(streams)
    dec.poison_output()
    dec.decode()
    for i, (d, s) in enumerate(zip(datas, streams)):
        rc, ref = oracle_decode(s)
        assert rc == 0 and np.array_equal(ref, d), chunk + i
        assert np.array_equal(dec.output(i), ref), (chunk + i, random_plain(chunk + i)[1])


@pytest.mark.parametrize("seed", range(0, N_PLAIN, 10))
def test_random_streams_single_asset_entry(api, seed):
    """The host-pointer entry (DecodeGPU), one stream per call."""
    d, kw = random_plain(seed)
    s = E.encode(d, **kw)
    out, _ = api.DecodeGPU(s)
    assert np.array_equal(out, d), (seed, kw)


def test_random_preconditioned_streams(api, kernel_form):
    texs, streams = [], []
    for seed in range(N_PRECON):
        tex, pre, kw = random_precon(seed)
        texs.append(tex)
        streams.append(E.encode(tex, precondition=pre, **kw))
    dec = api.BatchDecoder(streams, out_sizes=[len(t) for t in texs])
    dec.poison_output()
    dec.decode()
    for i, (t, s) in enumerate(zip(texs, streams)):
        rc, ref = oracle_decode(s, out_size=len(t))
        assert rc == 0 and np.array_equal(ref, t), i
        assert np.array_equal(dec.output(i), ref), (i, random_precon(i)[1])


def test_soak_batch_260800_decondition_tile_rotation(api):
    """Regression (round 5 soak): 80 valid pre-conditioned streams; stream 66 (BC1, 71 x 14 blocks, 2 mips = 11 super-tiles) made the
    de-conditioning kernel's tile rotation `(hash >> 8) % 11` come out as 0xFFFFFF on the device -- a 24-bit remainder lowered to one float
    reciprocal, one too large in the quotient -- and the kernel walked off its table: GPU memory access fault (the simulator divides exactly).
    The rotation is a multiply-high now (tests/test_kernel_isa.py keeps the lowering out of the kernels)."""
    texs, streams = [], []
    for seed in range(260800, 260880):
        tex, pre, kw = random_precon(seed)
        texs.append((tex, pre))
        streams.append(E.encode(tex, precondition=pre, **kw))
    dec = api.BatchDecoder(streams, out_sizes=[len(t) for t, _ in texs])
    dec.poison_output()
    dec.decode()
    for i, ((t, pre), s) in enumerate(zip(texs, streams)):
        rc, ref = oracle_decode(s, out_size=len(t))
        assert rc == 0 and np.array_equal(dec.output(i), ref), (260800 + i, pre)


def _valid_reference(api):
    d, kw = random_plain(3)
    s = E.encode(d, **kw)
    return d, s


def test_corrupt_streams_return_a_status_and_leave_the_device_usable(api):
    """120 damaged streams (bit flips, truncation, page-table / page-header / precondition-header damage), decoded
    alone and in batches next to valid streams.  Any status is acceptable, a fault is not: the valid neighbours
    in the same launch and a valid decode afterwards must still be bit-exact."""
    import torch
    good_d, good_s = _valid_reference(api)
    n_bad_status = 0
    for base in range(0, 120, 12):
        streams, sizes, expect = [], [], []
        for seed in range(base, base + 12):
            if seed % 3 == 1:
                tex, pre, kw = random_precon(seed)
                s, n = E.encode(tex, precondition=pre, **kw), len(tex)
            else:
                d, kw = random_plain(seed)
                s, n = E.encode(d, **kw), len(d)
            bad, kind = corrupt(s, seed)
            if bad[0] != 5 or bad[1] != 250:                        # the two header checks are covered by test_error_codes
                continue
            streams.append(bad); sizes.append(n); expect.append(None)
            streams.append(good_s); sizes.append(len(good_d)); expect.append(good_d)
        dec = api.BatchDecoder(streams, out_sizes=sizes)
        dec.poison_output()
        try:
            dec.decode()
        except api.BrotligError as e:
            assert e.code in (api.BROTLIG_ERROR_CORRUPT_STREAM, api.BROTLIG_ERROR_GENERIC)
            n_bad_status += 1
        torch.cuda.synchronize()
        for i, exp in enumerate(expect):
            if exp is not None:
                assert np.array_equal(dec.output(i), exp), (base, i)
        guard = dec.d_out[dec.out_bytes:].cpu().numpy()
        assert np.all(guard == 0xCD), "a damaged stream wrote past the output buffer"
        del dec
    assert n_bad_status > 0                                         # the damage is noticed at least sometimes
    out, _ = api.DecodeGPU(good_s)
    assert np.array_equal(out, good_d)


def test_truncated_and_damaged_single_assets(api):
    """DecodeGPU on damaged single streams: an error code or wrong bytes, never a fault; then a valid decode."""
    good_d, good_s = _valid_reference(api)
    for seed in range(200, 240):
        d, kw = random_plain(seed)
        s = E.encode(d, **kw)
        bad, kind = corrupt(s, seed)
        try:
            api.DecodeGPU(bad, output_size=len(d))
        except api.BrotligError as e:
            assert e.code in (api.BROTLIG_ERROR_CORRUPT_STREAM, api.BROTLIG_ERROR_INCORRECT_STREAM_FORMAT, api.BROTLIG_ERROR_GENERIC)
    out, _ = api.DecodeGPU(good_s)
    assert np.array_equal(out, good_d)


def test_damaged_distance_below_zero_in_either_kernel(api):
    """Five damaged streams from the device soak of round 4 (a ring code applied to a small distance wraps below zero): the two-wavefront
    kernel handed the wrapped distance to its consumer with the flags OR-ed on top -- a memory access fault on the device.  Both kernels
    must refuse the copy and report a bad page; a valid stream in the same launch stays bit-exact."""
    good_d, good_s = _valid_reference(api)
    streams, sizes = [], []
    for seed in (150069, 150248, 150466, 151219, 151529):
        d, kw = random_plain(seed)
        bad, kind = corrupt(E.encode(d, **kw), seed)
        streams.append(bad); sizes.append(len(d))
    streams.append(good_s); sizes.append(len(good_d))
    for mode in (0, 1, 2):
        api.DebugSetDecodeMode(mode)
        try:
            dec = api.BatchDecoder(streams, out_sizes=sizes)
            dec.poison_output()
            with pytest.raises(api.BrotligError):
                dec.decode()
            assert np.array_equal(dec.output(len(streams) - 1), good_d), mode
        finally:
            api.DebugSetDecodeMode(0)
    out, _ = api.DecodeGPU(good_s)
    assert np.array_equal(out, good_d)


def test_empty_page_behind_a_damaged_table_entry_is_rejected(api):
    """Batch 150080 of round 4's device soak (forty damaged streams, output sizes as their headers claim, a valid stream last): one of
    them describes a page of zero bytes outside its stream, which fetch_job let through -- a memory access fault in either kernel.  Any
    status, no fault, the valid neighbour bit-exact, in all three kernel modes."""
    streams, good = [], None
    for seed in range(150080, 150120):
        d, kw = random_plain(seed)
        bad, kind = corrupt(E.encode(d, **kw), seed)
        streams.append(bad)
    good, kw = random_plain(3)
    streams.append(E.encode(good, **kw))
    for mode in (0, 1, 2):
        api.DebugSetDecodeMode(mode)
        try:
            try:
                dec = api.BatchDecoder(streams, out_sizes=None)
            except api.BrotligError:
                continue
            dec.poison_output()
            try:
                dec.decode()
            except api.BrotligError:
                pass
            assert np.array_equal(dec.output(len(streams) - 1), good), mode
        finally:
            api.DebugSetDecodeMode(0)


def test_undersized_output_buffer_is_refused_for_preconditioned_streams(api):
    """the de-conditioning kernel writes the whole texture; with out_bytes smaller than the texture the
    stream must be rejected by the prepare kernel (status), and nothing may be written past out_bytes."""
    import torch
    from brotli_g_sdk_amd import datagen as D
    tex = D.bc_texture(3, 128, 96, seed=11)                                # 192 KiB
    s = E.encode(tex, precondition=dict(format=3, width_blocks=128, height_blocks=96, swizzle=1, delta=1))
    dec = api.BatchDecoder([s], out_sizes=[len(tex)])
    dec.poison_output()
    short = len(tex) // 2
    args = dec._args(dec._stream())
    args[3] = short                                                  # lie about the output capacity
    with torch.cuda.device(dec.device):
        rc = api.lib().BrotligDecodeBatchDevice(*args)
        assert rc == 0
        rc = api.lib().BrotligDecodeBatchStatus(dec.d_ws.data_ptr(), dec._stream())
    assert rc != 0
    torch.cuda.synchronize()
    assert bool((dec.d_out[short:] == 0xCD).all())


def test_damaged_header_cannot_reach_a_neighbouring_stream(api):
    """A stream whose header claims more output than its descriptor's out_capacity (here: LastPageSize with bit 17
    set, i.e. a "short" last page of 128 KiB + 505 bytes in a 64 KiB page) is rejected as a whole; the stream
    laid out right behind it decodes bit-exactly."""
    d = np.frombuffer(bytes(range(256)) * 2, dtype=np.uint8)[:505].copy()
    s = E.encode(d)
    bad = s.copy()
    bad[6] |= 0x08
    good_d, good_s = _valid_reference(api)
    dec = api.BatchDecoder([bad, good_s], out_sizes=[len(d), len(good_d)])
    dec.poison_output()
    with pytest.raises(api.BrotligError):
        dec.decode()
    assert np.array_equal(dec.output(1), good_d)
