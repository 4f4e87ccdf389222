"""CPU model of the corpus-tile sequence of ip_topk_fast_kernel (csrc/ip_topk_fast.hip): every workgroup
(query tile, split) walks the windows in order and inside a window the Ws tiles of its split.  The model
restates the kernel's index arithmetic line by line and checks what the kernel relies on: the splits of one
query tile cover every corpus tile exactly once, a workgroup arrives at every window boundary that is not
the last one exactly once, and only workgroups that still have work wait there."""
import pytest


def tile_sequence(n_tiles, S, Ws, split):
    """(tiles visited, boundaries arrived at, boundaries waited at) of one workgroup -- the loop of the kernel."""
    W = Ws * S
    n_win = (n_tiles + W - 1) // W
    t, jw, win = split * Ws, 0, 0
    have = t < n_tiles
    tiles, arrived, waited = [], [], []
    while have:
        jn, winn = jw + 1, win
        if jn < Ws:
            tn = t + 1
        else:
            jn, winn = 0, win + 1
            tn = winn * W + split * Ws
        have_n = tn < n_tiles
        tiles.append(t)
        if winn != win and winn < n_win:
            arrived.append(win)
            if have_n:
                waited.append(win)
        t, jw, win, have = tn, jn, winn, have_n
    return tiles, arrived, waited


@pytest.mark.parametrize("S", [1, 2, 4, 8, 32])
@pytest.mark.parametrize("Ws", [1, 3, 16, 128])
def test_every_tile_once_and_boundaries(S, Ws):
    W = S * Ws
    for n_tiles in sorted({1, 2, S, S + 1, W - 1, W, W + 1, 2 * W, 2 * W + Ws, 3 * W - 1, 5 * W + 7, 1000}):
        if n_tiles < 1:
            continue
        n_win = (n_tiles + W - 1) // W
        seen = []
        for split in range(S):
            tiles, arrived, waited = tile_sequence(n_tiles, S, Ws, split)
            seen += tiles
            assert all(0 <= t < n_tiles for t in tiles)
            assert tiles == sorted(tiles)  # ascending image rows per workgroup
            # arrives exactly once at every boundary before the last window (all of those windows are full)
            assert arrived == list(range(n_win - 1)), (n_tiles, S, Ws, split)
            assert set(waited) <= set(arrived)
        assert sorted(seen) == list(range(n_tiles)), (n_tiles, S, Ws)


def test_host_window_choice_covers_small_shards():
    """make_fast_plan: Wt = min(window, n_tiles); Ws = ceil(Wt / S) -- with one window every split gets a share."""
    for n_tiles in (16, 17, 100, 255, 256, 257):
        for S in (1, 2, 4, 8):
            if S * 8 > n_tiles:
                continue
            Wt = min(256, n_tiles)
            Ws = (Wt + S - 1) // S
            counts = [len(tile_sequence(n_tiles, S, Ws, s)[0]) for s in range(S)]
            assert sum(counts) == n_tiles
            if n_tiles <= 256:
                assert max(counts) - min(counts) <= Ws  # single window: contiguous shares
