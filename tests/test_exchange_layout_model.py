"""CPU model of the recurrence's exchange-buffer addressing (csrc/recurrent.hip: brnn_recurrent_q_kernel,
brnn_recurrent_t_kernel and brnn_recurrent_kernel<NTW>; lane_order_steps): a tile of 16 utterances keeps its KB per
16-unit chunk in LANE order ([k quarter][utterance][4 units]: lane l = uj + 16 kq at byte 16 l) while its 16 exchange
rows lie inside the step's block, and row-major ([utterance][16 units]: lane at 64 uj + 16 kq) after that.  Round 6: the
engine (csrc/brnn_engine.hip plan_minibatch, recurrent.h recurrent_step_xrows) rounds a step's block up to 16 rows from
17 alive utterances on (to 4 below), so every tile but the first keeps lane order for its whole life and the first one
while 13 or more utterances are alive (round 5: blocks of 4 rows, lane order only while all 16 utterances of the tile
were alive -- ragged minibatches ran mostly row-major).  The kernels' claim, checked here on ragged minibatches: what a
lane reads at step j + 1 is what the same (utterance, k quarter) wrote at step j, no two lanes write the same bytes, and
nothing is written outside the step's own block.  (Reference context: the state h_{t-1} that nnets/brnnet.py:120-135
multiplies with the recurrent weights.)"""
import numpy as np
import pytest


def step_blocks(Ts):
    """xbase[j] and the active count per step, utterances sorted longest first (brnn_engine.hip, plan_minibatch)"""
    Ts = sorted(Ts, reverse=True)
    xbase, nact, xb = [], [], 0
    for j in range(Ts[0]):
        na = sum(1 for t in Ts if t > j)
        xbase.append(xb)
        nact.append(na)
        xb += step_rows(na)
    return Ts, xbase, nact, xb


def step_rows(na):
    """recurrent.h recurrent_step_xrows"""
    return (na + 15) & ~15 if na >= 17 else (na + 3) & ~3


def lane_order_steps(Ts, tile, b_off=0):
    """recurrent.hip lane_order_steps: Ts the launch's sorted lengths"""
    if tile == 0 and b_off == 0:
        return Ts[12] if len(Ts) > 12 else 0
    return Ts[16 * tile] if 16 * tile < len(Ts) else 0


def lane_offset(xb, tile, lane, uT, tile_T, j, lane_order_allowed=True):
    """byte offset (within a chunk) that lane (uj = lane % 16, kq = lane // 16) of `tile` uses for step j's rows"""
    uj, kq = lane % 16, lane // 16
    if lane_order_allowed and j < tile_T:
        return (xb + tile * 16) * 64 + lane * 16
    return (xb + tile * 16 + uj) * 64 + kq * 16


def consumer_offset(xb_prev, tile, lane, tile_T, jc):
    """where the kernel's lane reads the previous step's state at step jc > 0 (`if (j > 0 && j <= tile_T && active)`)"""
    uj, kq = lane % 16, lane // 16
    if jc <= tile_T:
        return (xb_prev + tile * 16) * 64 + lane * 16
    return (xb_prev + tile * 16 + uj) * 64 + kq * 16


@pytest.mark.parametrize("B,seed", [(16, 0), (17, 1), (32, 2), (24, 3), (64, 4), (100, 5), (128, 6), (5, 7)])
def test_lane_order_tiles_round_trip_and_stay_inside_their_block(B, seed):
    rs = np.random.RandomState(seed)
    Ts, xbase, nact, n_rows = step_blocks([int(t) for t in rs.randint(1, 20, size=B)])
    ntiles = (B + 15) // 16
    tile_T = [lane_order_steps(Ts, t) for t in range(ntiles)]
    for j in range(Ts[0]):
        owner = {}
        lo, hi = xbase[j] * 64, (xbase[j] + step_rows(nact[j])) * 64
        for tile in range(ntiles):
            for lane in range(64):
                ub = tile * 16 + lane % 16
                if ub >= B or j >= Ts[ub]:
                    continue                                   # finished / empty slots store nothing
                off = lane_offset(xbase[j], tile, lane, Ts[ub], tile_T[tile], j)
                assert lo <= off and off + 16 <= hi, "write outside the step's block"
                for q in range(off, off + 16, 4):
                    assert q not in owner, "two lanes write the same dword"
                    owner[q] = (ub, lane // 16, (q - off) // 4)
        if j + 1 >= Ts[0]:
            continue
        for tile in range(ntiles):                             # the consumer of step j + 1 (kernel: `j <= tile_T`)
            for lane in range(64):
                ub = tile * 16 + lane % 16
                if ub >= B or j + 1 >= Ts[ub]:
                    continue
                off = consumer_offset(xbase[j], tile, lane, tile_T[tile], j + 1)
                for q in range(off, off + 16, 4):
                    assert owner[q] == (ub, lane // 16, (q - off) // 4)


def test_lane_order_needs_the_tiles_sixteen_rows_inside_the_block():
    """the reason for the rule: 20 utterances alive -> round 5's block of 20 rows; tile 1 in lane order would spread its
    four utterances over 16 rows' worth of bytes, 12 of them the next step's block.  Round 6 rounds that block up to 32
    rows: tile 1 fits.  Tile 0 with 9 utterances alive has a block of 12 rows: lane order would still leave it."""
    xb = 0
    worst_tile1 = max((xb + 16) * 64 + lane * 16 + 16 for lane in range(64) if lane % 16 < 4)
    assert worst_tile1 > (xb + ((20 + 3) & ~3)) * 64            # round 5's block
    assert worst_tile1 <= (xb + step_rows(20)) * 64             # round 6's
    worst_tile0 = max(xb * 64 + lane * 16 + 16 for lane in range(64) if lane % 16 < 9)
    assert worst_tile0 > (xb + step_rows(9)) * 64
    assert step_rows(13) == 16 and step_rows(12) == 12 and step_rows(17) == 32 and step_rows(16) == 16


@pytest.mark.parametrize("B,b0,seed", [(48, 32, 0), (80, 64, 1), (96, 64, 2), (130, 128, 3)])
def test_later_launches_of_a_cut_minibatch_keep_lane_order_throughout(B, b0, seed):
    """a minibatch cut into launches (recurrent.hip launch_recurrent: utterances b0.. run as a launch of their own, with
    LOCAL utterance indices on the minibatch's step blocks): whenever one of the launch's utterances is alive, more than
    16 of the minibatch are, so its tiles' rows are inside the block for their whole life"""
    rs = np.random.RandomState(seed)
    Ts, xbase, nact, _ = step_blocks([int(t) for t in rs.randint(1, 20, size=B)])
    local = Ts[b0:]
    for tile in range((len(local) + 15) // 16):
        steps = lane_order_steps(local, tile, b_off=b0)
        assert steps == local[16 * tile]
        for j in range(steps):
            assert (tile + 1) * 16 <= step_rows(nact[j])


def test_a_wave_access_is_one_contiguous_kb_in_lane_order_and_eight_split_lines_row_major():
    lane_order = sorted(lane_offset(0, 0, lane, 9, 9, 0) for lane in range(64))
    assert lane_order == list(range(0, 1024, 16))
    for group in range(4):                                     # a 16-lane group of a dwordx4 access
        lanes = range(16 * group, 16 * group + 16)
        lines_lane = {lane_offset(0, 0, lane, 9, 9, 0) // 128 for lane in lanes}
        lines_row = {lane_offset(0, 0, lane, 9, 9, 0, lane_order_allowed=False) // 128 for lane in lanes}
        assert len(lines_lane) == 2 and len(lines_row) == 8
