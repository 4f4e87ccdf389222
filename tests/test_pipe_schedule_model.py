"""Model check of the ping-pong main-loop schedules of ance_amd/csrc/pipe256.h (CPU only).

The kernels keep LDS-DMA loads in flight across barriers with counted ``s_waitcnt vmcnt(N)`` and let two
wave groups run one barrier apart, so correctness rests on a hazard argument (header of pipe256.h).
This test restates the schedule tables -- what each phase reads, stages and waits for, in the prologue,
the steady state and the two peeled tail tiles -- and replays them on a slot timeline for both wave
groups, asserting for every K-tile count:

  RAW  a half-tile is read only after BOTH groups retired their pieces of it with a wait that lies in
       an earlier slot (a barrier separates slots), and it is the half-tile of the right K-tile;
  WAR  a buffer is restaged only after both groups' reads of its previous occupant have completed
       (a read completes in the MFMA half-phase that consumes it) in an earlier slot;
  END  nothing is in flight when the loop ends, and every half-tile of every K-tile was read.

vmcnt semantics: each wave's loads retire in issue order; ``wait(N)`` returns when at most N are
outstanding.  A staged half-tile is two pieces per wave, so its count is 2.
"""
import pytest

A0, A1, B0, B1 = 0, 1, 2, 3


def four_phase(mode, t, keep_b0):
    """Phases of K-tile t: (reads, wait or None, stages) -- pipe256.h ``tile<MODE>``."""
    c3_reads = [] if keep_b0 else [(B0, t)]
    return [
        ([(A0, t), (B0, t)], None, [(B0, t + 1)] if mode <= 1 else []),
        ([(B1, t)], None, [(A0, t + 2)] if mode == 0 else []),
        ([(A1, t)], None, [(B1, t + 2)] if mode == 0 else []),
        (c3_reads, {0: 4, 1: 0, 2: None}[mode], [(A1, t + 2)] if mode == 0 else []),
    ]


def four_phase_prologue():
    # stage<0>(0) stage<2>(0) stage<3>(0) stage<1>(0) stage<0>(1) stage<3>(1) stage<1>(1); vmcnt(6); barrier
    return [[(A0, 0), (B0, 0), (B1, 0), (A1, 0), (A0, 1), (B1, 1), (A1, 1)], 6, []]


def coarse(mode, t):
    """pipe256.h ``tile2<MODE>``."""
    return [
        ([(A0, t), (B0, t), (B1, t)], 6 if mode <= 1 else 0, [(A1, t + 1)] if mode <= 1 else []),
        ([(A1, t)], 2 if mode <= 1 else None, [(A0, t + 2), (B0, t + 2), (B1, t + 2)] if mode == 0 else []),
    ]


def coarse_pair3_early(form):
    """pipe256.h ``tile2<MODE>`` with PIPE_PAIR3_EARLY = form (1 or 2): phases carry a fourth entry, the half-tiles staged in the
    READ half-phase (after the reads, before the wait), and these forms wait for their own ds_reads before the barrier."""
    def phases(mode, t):
        p0 = ([(A0, t), (B0, t), (B1, t)], 8 if mode <= 1 else 0, [], [(A1, t + 1)] if mode <= 1 else [])
        if mode == 0:
            if form == 2:
                p1 = ([(A1, t)], 8, [], [(A0, t + 2), (B0, t + 2), (B1, t + 2)])
            else:
                p1 = ([(A1, t)], 4, [(B0, t + 2), (B1, t + 2)], [(A0, t + 2)])
        else:
            p1 = ([(A1, t)], 2 if mode == 1 else None, [], [])
        return [p0, p1]
    return phases


def coarse_prologue():
    # stage tile 0 (A0 B0 B1 A1); vmcnt(2); barrier; stage A0 B0 B1 of tile 1
    return [[(A0, 0), (B0, 0), (B1, 0), (A1, 0)], 2, [(A0, 1), (B0, 1), (B1, 1)]]


def replay(nk, phases_of, prologue, reads_complete_before_barrier=False):
    """Replays a loop of ``nk`` K-tiles.  reads_complete_before_barrier: the schedule waits ``lgkmcnt(0)`` before the barrier that
    ends a read half-phase, so a read has completed in that slot (otherwise: in the MFMA half-phase that consumes it).  (The search filter's streamed loop over several corpus tiles is, from the
    schedule's point of view, one longer loop: K-tile indices simply continue into the next corpus tile.)"""
    total = nk
    phases = []
    for t in range(total):
        mode = 0 if t < nk - 2 else (1 if t == nk - 2 else 2)
        phases += phases_of(mode, t)
    # per group: FIFO of outstanding (half_tile, tile), and the slot in which each half-tile was retired
    fifo = {0: [], 1: []}
    retired = {0: {}, 1: {}}       # (type, tile) -> slot of the wait that retired it
    issued = {0: {}, 1: {}}        # (type, tile) -> slot of issue
    read_done = {0: {}, 1: {}}     # (type, tile) -> slot in which the read was consumed

    def wait_pieces(g, n_pieces, slot):
        assert n_pieces % 2 == 0  # the fifo holds half-tiles, two pieces each
        while len(fifo[g]) > n_pieces // 2:
            retired[g][fifo[g].pop(0)] = slot

    def stage(g, ht, slot):
        typ, tile = ht
        prev = (typ, tile - 2)
        # WAR: both groups finished reading the previous occupant of this buffer in an earlier slot
        if tile >= 2:
            for gg in (0, 1):
                assert prev in read_done[gg], "restage of %s before group %d read %s" % (ht, gg, prev)
                assert read_done[gg][prev] < slot, "WAR: %s restaged in slot %d, group %d consumed %s in slot %d" % (
                    ht, slot, gg, prev, read_done[gg][prev])
        fifo[g].append(ht)
        issued[g][ht] = slot

    # prologue: both groups together in slot -1 (then the prologue barrier; then group 1's extra barrier)
    pro_first, pro_wait, pro_second = prologue
    for g in (0, 1):
        for ht in pro_first:
            stage(g, ht, -1)
        wait_pieces(g, pro_wait, -1)
        for ht in pro_second:
            stage(g, ht, -1 if g == 0 else 0)  # issued after the prologue barrier, before the group's first phase
    n_slots = 2 * len(phases) + 2
    for slot in range(n_slots):
        for g in (0, 1):
            # group g: R(p) in slot 2p + g, M(p) in slot 2p + 1 + g
            if (slot - g) % 2 == 0 and 0 <= (slot - g) // 2 < len(phases):
                ph = phases[(slot - g) // 2]
                reads, w = ph[0], ph[1]
                early = ph[3] if len(ph) > 3 else []
                for ht in reads:
                    # RAW: retired by both groups in an earlier slot
                    for gg in (0, 1):
                        assert ht in retired[gg], "read of %s in slot %d: group %d never waited for it" % (ht, slot, gg)
                        assert retired[gg][ht] < slot, "RAW: %s read in slot %d, group %d retired it in slot %d" % (
                            ht, slot, gg, retired[gg][ht])
                for ht in early:  # LDS-DMAs issued in the read half-phase, behind the reads
                    if ht[1] < total:
                        stage(g, ht, slot)
                if w is not None:
                    wait_pieces(g, w, slot)
                if reads_complete_before_barrier:
                    for ht in reads:
                        read_done[g][ht] = slot
            if (slot - g) % 2 == 1 and 0 <= (slot - g - 1) // 2 < len(phases):
                p = (slot - g - 1) // 2
                reads, stages = phases[p][0], phases[p][2]
                for ht in reads:
                    read_done[g].setdefault(ht, slot)  # consumed by this half-phase's MFMAs
                for ht in stages:
                    if ht[1] < total:
                        stage(g, ht, slot)
    for g in (0, 1):
        assert not fifo[g], "loads left in flight: %s" % fifo[g]
        for t in range(total):
            for typ in (A0, A1, B0, B1):
                assert (typ, t) in read_done[g], "half-tile %s of tile %d never read by group %d" % (typ, t, g)


@pytest.mark.parametrize("nk", range(2, 14))
@pytest.mark.parametrize("keep_b0", [False, True])
def test_four_phase_schedule(nk, keep_b0):
    replay(nk, lambda mode, t: four_phase(mode, t, keep_b0), four_phase_prologue())


@pytest.mark.parametrize("nk", range(2, 14))
def test_coarse_schedule(nk):
    replay(nk, coarse, coarse_prologue())


@pytest.mark.parametrize("nk", range(2, 14))
@pytest.mark.parametrize("form", [1, 2])
def test_coarse_pair3_early_schedules(nk, form):
    """The split GEMM's forms with LDS-DMAs issued in the read half-phases (PIPE_PAIR3_EARLY)."""
    replay(nk, coarse_pair3_early(form), coarse_prologue(), reads_complete_before_barrier=True)
    with pytest.raises(AssertionError):  # without the lgkmcnt(0) before the barrier the early restage races the other group's reads
        replay(max(nk, 4), coarse_pair3_early(form), coarse_prologue(), reads_complete_before_barrier=False)


def test_model_catches_a_broken_schedule():
    """The checker is not vacuous: restaging one phase earlier than allowed, or waiting for too little, fails."""
    def early_restage(mode, t):
        ph = four_phase(mode, t, False)
        if mode == 0:  # stage A-half0 of tile t+2 already in phase c0 (its buffer is read in c0 of tile t)
            ph[0] = (ph[0][0], ph[0][1], ph[0][2] + [(A0, t + 2)])
            ph[1] = (ph[1][0], ph[1][1], [])
        return ph
    with pytest.raises(AssertionError):
        replay(8, early_restage, four_phase_prologue())

    def lazy_wait(mode, t):
        ph = coarse(mode, t)
        if mode == 0:
            ph[1] = (ph[1][0], 4, ph[1][2])  # vmcnt(4) instead of vmcnt(2): B-half1 of the next tile may still be in flight
        return ph
    with pytest.raises(AssertionError):
        replay(8, lazy_wait, coarse_prologue())


# ---- persistent streaming GEMM (gemm256_f16.hip: gemm256_split_stream_kernel) -----------------------------------------------
def replay_persistent(nk, n_out, f_actual, f_claimed, extra_ops_group1=1):
    """A workgroup of the persistent split GEMM: ``n_out`` output tiles of ``nk`` K-tiles each on the coarse schedule.  The K loop
    of output tile o hands over to tile o + 1 (its last two K-tiles stage K-tiles 0 and 1 of the next tile); between two tiles
    both wave groups are aligned (leave()), run the epilogue -- which issues ``f_actual`` vector-memory operations per wave
    (group 1: ``extra_ops_group1`` more, the waves that also fetch the parameter vectors) and uses the A-half1 slot of stage
    buffer 1 as slab space -- and enter() the next K loop, whose K-tile 0 waits with 6 + f_claimed / 2 + f_claimed (capped at 63).
    The first tile starts from prologue_landed (nothing in flight).  Asserts RAW / WAR as ``replay`` and that the slab slot is
    neither still being read nor restaged while the epilogue owns it."""
    assert nk % 2 == 0 and nk >= 4  # the hand-over keeps the buffer parity of K-tile 0 only for an even count
    vm0 = min(63, 6 + f_claimed)
    vm1 = min(63, 2 + f_claimed)
    fifo = {0: [], 1: []}          # [(item, pieces)], oldest first; item = (type, global K-tile) or ("x", n)
    retired = {0: {}, 1: {}}
    read_done = {0: {}, 1: {}}
    slab_busy = []                  # slots in which the epilogue owns (A1, buffer 1)
    total = nk * n_out

    def wait_pieces(g, n, slot):
        out = sum(p for _, p in fifo[g])
        r = out - n                 # pieces that must retire, oldest first (vmcnt retires in order)
        while r > 0 and fifo[g]:
            item, p = fifo[g][0]
            if p > r:
                fifo[g][0] = (item, p - r)   # part of a half-tile's pieces: the half-tile itself is not retired yet
                r = 0
            else:
                fifo[g].pop(0)
                r -= p
                if item[0] != "x":
                    retired[g][item] = slot

    def stage(g, ht, slot):
        typ, gt = ht
        prev = (typ, gt - 2)
        if gt >= 2:
            for gg in (0, 1):
                assert prev in read_done[gg] and read_done[gg][prev] < slot, "WAR: %s restaged in slot %d" % (ht, slot)
        if typ == A1 and gt % 2 == 1:
            assert slot not in slab_busy, "A-half1 slot of buffer 1 restaged while the epilogue's slabs live there"
        fifo[g].append((ht, 2))

    for g in (0, 1):  # prologue_landed
        for ht in [(A0, 0), (B0, 0), (B1, 0), (A1, 0), (A0, 1), (B0, 1), (B1, 1)]:
            retired[g][ht] = -1
    base = 0
    for o in range(n_out):
        phases = []
        for t in range(nk):
            gt = o * nk + t
            last = o == n_out - 1
            mode = 0 if not (last and t >= nk - 2) else (1 if t == nk - 2 else 2)
            p0, p1 = coarse(mode, gt)
            if t == 0:
                p0 = (p0[0], vm0, p0[2])
                p1 = (p1[0], vm1, p1[2])
            phases += [p0, p1]
        for ls in range(2 * len(phases) + 2):
            slot = base + ls
            for g in (0, 1):
                if (ls - g) % 2 == 0 and 0 <= (ls - g) // 2 < len(phases):
                    reads, w = phases[(ls - g) // 2][0], phases[(ls - g) // 2][1]
                    for ht in reads:
                        for gg in (0, 1):
                            assert ht in retired[gg] and retired[gg][ht] < slot, "RAW: %s read in slot %d (group %d: %s)" % (
                                ht, slot, gg, retired[gg].get(ht))
                    if w is not None:
                        wait_pieces(g, w, slot)
                if (ls - g) % 2 == 1 and 0 <= (ls - g - 1) // 2 < len(phases):
                    reads, stages = phases[(ls - g - 1) // 2][0], phases[(ls - g - 1) // 2][2]
                    for ht in reads:
                        read_done[g].setdefault(ht, slot)
                    for ht in stages:
                        if ht[1] < total:
                            stage(g, ht, slot)
        base += 2 * len(phases) + 2
        if o < n_out - 1:
            # epilogue: its own slot (leave() aligned the groups; a barrier follows it)
            for g in (0, 1):
                assert read_done[g][(A1, o * nk + nk - 1)] < base, "slab written while A-half1 of the last K-tile is still read"
                fifo[g] += [(("x", i), 1) for i in range(f_actual + (extra_ops_group1 if g == 1 else 0))]
            slab_busy.append(base)
            base += 1
    for g in (0, 1):
        assert not [it for it, _ in fifo[g] if it[0] != "x"], "LDS-DMAs left in flight: %s" % fifo[g]
        for gt in range(total):
            for typ in (A0, A1, B0, B1):
                assert (typ, gt) in read_done[g]


@pytest.mark.parametrize("nk", [4, 6, 8, 24])
@pytest.mark.parametrize("n_out", [1, 2, 3])
@pytest.mark.parametrize("f", [0, 35])
def test_persistent_gemm_hand_over(nk, n_out, f):
    """f = the foreign operations of gemm256_f16.hip: EPS_FOREIGN_OPS (32 stores + 3 LDS-DMAs for QKV and GELU alike); 0 = the
    steady-state waits on K-tile 0 (ANCE_STREAM_LOOSE_FIRST=0: correct whatever the epilogue issues)."""
    replay_persistent(nk, n_out, f_actual=f, f_claimed=f)
    replay_persistent(nk, n_out, f_actual=f + 20, f_claimed=f)   # an epilogue that issues MORE than claimed only waits longer
    replay_persistent(nk, n_out, f_actual=max(f, 5), f_claimed=0)


def test_persistent_model_catches_an_overstated_count():
    """Claiming more foreign operations than every wave really issues lets K-tile 0 read a half-tile that is still in flight."""
    with pytest.raises(AssertionError):
        replay_persistent(8, 2, f_actual=30, f_claimed=35)
    with pytest.raises(AssertionError):
        replay_persistent(8, 3, f_actual=0, f_claimed=3, extra_ops_group1=0)
