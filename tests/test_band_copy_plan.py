"""CPU: the copies that deliver a device's bands of a multi-device context (rt_api.hip deliver_bands, reference seam:
the read-back at skeleton.cpp:179-180).  The plan is what rt_render / rt_render_device enqueue; the cross-device branches
(peer mapping present / absent) cannot execute on a one-GPU box, so the plan itself is checked: exact (op, offsets,
pitches, sizes) tuples for ragged heights, and a replay with host memcpy must assemble the frame."""
import numpy as np
import pytest

from uob_raytracer_amd import abi, bands, runtime as rt


def replay(plan, stripe_bytes, dst_bytes):
    for q in plan:
        if q["op"] == abi.RT_COPY_2D:
            for r in range(q["rows"]):
                s0, d0 = q["src_offset"] + r * q["src_pitch"], q["dst_offset"] + r * q["dst_pitch"]
                dst_bytes[d0:d0 + q["width_bytes"]] = stripe_bytes[s0:s0 + q["width_bytes"]]
        else:
            assert q["rows"] == 1
            dst_bytes[q["dst_offset"]:q["dst_offset"] + q["width_bytes"]] = stripe_bytes[q["src_offset"]:q["src_offset"] + q["width_bytes"]]


@pytest.mark.parametrize("elem", [4, 16])
@pytest.mark.parametrize("N,dbr,W,H", [(2, 32, 64, 256), (3, 8, 50, 173), (4, 16, 33, 20), (8, 32, 40, 4096 // 8 + 5), (5, 7, 9, 7), (2, 32, 16, 31)])
@pytest.mark.parametrize("dev_to_dev,peer_ok", [(0, 1), (1, 1), (1, 0)])
def test_plan_assembles_the_frame(N, dbr, W, H, elem, dev_to_dev, peer_ok):
    rng = np.random.default_rng(N * 1000 + H)
    frame = rng.integers(0, 2 ** 32, size=(H, W * elem // 4), dtype=np.uint32)         # what the single device would produce
    out = np.zeros(H * W * elem, np.uint8)
    for k in range(N):
        rows = bands.band_rows_of(k, N, H, dbr)
        stripe = frame[rows].copy().view(np.uint8).ravel() if rows else np.zeros(0, np.uint8)
        plan = rt.band_copy_plan(N, k, dbr, W, H, elem, dev_to_dev, peer_ok, same_device=(k == 0))
        for q in plan:      # every copy stays inside its buffers
            assert q["src_offset"] + (q["rows"] - 1) * q["src_pitch"] + q["width_bytes"] <= stripe.size
            assert q["dst_offset"] + (q["rows"] - 1) * q["dst_pitch"] + q["width_bytes"] <= out.size
        replay(plan, stripe, out)
    assert np.array_equal(out.view(np.uint32).reshape(frame.shape), frame)


def test_exact_tuples_ragged_height():
    """3 devices, bands of 8 rows, 50 x 173 ARGB frame: 173 = 21 bands + 5 rows; device 0 owns bands 0,3,..,21 (the ragged one is
    band 21: 8 full bands + 5 rows ... no: band 21 = rows 168..172 belongs to device 21 % 3 = 0)."""
    W, H, N, dbr, e = 50, 173, 3, 8, 4
    band = dbr * W * e
    # device 0: bands 0,3,...,18 full (7), band 21 ragged (5 rows)
    p0 = rt.band_copy_plan(N, 0, dbr, W, H, e, 1, 1, True)
    assert p0 == [dict(op=abi.RT_COPY_2D, dst_offset=0, dst_pitch=N * band, src_offset=0, src_pitch=band, width_bytes=band, rows=7),
                  dict(op=abi.RT_COPY_LINEAR, dst_offset=7 * N * band, dst_pitch=0, src_offset=7 * band, src_pitch=0, width_bytes=5 * W * e, rows=1)]
    # device 1 (another GPU, peer mapping present): bands 1,4,...,19 = 7 full bands, one 2-D copy over xGMI
    p1 = rt.band_copy_plan(N, 1, dbr, W, H, e, 1, 1, False)
    assert p1 == [dict(op=abi.RT_COPY_2D, dst_offset=band, dst_pitch=N * band, src_offset=0, src_pitch=band, width_bytes=band, rows=7)]
    # device 2 without a peer mapping: band by band through hipMemcpyPeerAsync
    p2 = rt.band_copy_plan(N, 2, dbr, W, H, e, 1, 0, False)
    assert p2 == [dict(op=abi.RT_COPY_PEER, dst_offset=2 * band + b * N * band, dst_pitch=0, src_offset=b * band, src_pitch=0,
                       width_bytes=band, rows=1) for b in range(7)]
    # the ragged band on another GPU travels as a peer copy whatever the mapping; to the host as a linear copy
    q = rt.band_copy_plan(2, 1, 32, 16, 31 + 32, 4, 1, 1, False)
    assert [c["op"] for c in q] == [abi.RT_COPY_PEER] and q[0]["width_bytes"] == 31 * 16 * 4 and q[0]["dst_offset"] == 32 * 16 * 4
    q = rt.band_copy_plan(2, 1, 32, 16, 31 + 32, 4, 0, 1, False)
    assert [c["op"] for c in q] == [abi.RT_COPY_LINEAR]


def test_invalid_arguments():
    with pytest.raises(rt.RtError):
        rt.band_copy_plan(0, 0, 32, 16, 16, 4, 0, 1, True)
    with pytest.raises(rt.RtError):
        rt.band_copy_plan(2, 2, 32, 16, 16, 4, 0, 1, True)
