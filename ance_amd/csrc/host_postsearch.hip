// Host-side post-search stage: hard-negative selection and the
// ann_training_data_N writer, as native code.  No GPU work happens here -- the reference does this
// stage in per-element Python (drivers/run_ann_data_gen.py:314-396), which becomes the tail of a
// refresh once the search takes seconds.
//
// Randomness contract: the reference draws one ``random.shuffle(list(range(k)))`` per effective
// query in row order and then one ``random.shuffle(list(range(n_rows)))`` for the line order.  To
// stay comparable with a seeded reference run these functions continue CPython's Mersenne Twister
// stream in place: the caller passes ``random.getstate()[1]`` (624 words + index) and installs the
// advanced state afterwards.  CPython semantics restated here:
//   shuffle(x):     for i in reversed(range(1, len(x))): j = _randbelow(i + 1); swap(x[i], x[j])
//   _randbelow(n):  k = n.bit_length(); r = getrandbits(k); while r >= n: r = getrandbits(k)
//   getrandbits(k): genrand_uint32() >> (32 - k)                               (k <= 32)
//   genrand_uint32: MT19937 (Matsumoto & Nishimura 2002), tempering 11 / 7,0x9d2c5680 / 15,0xefc60000 / 18
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ance_amd.h"
#include "common.h"

namespace ance {

namespace {

struct PyMT {
    uint32_t *mt;  // 624 words, caller-owned
    int index;

    explicit PyMT(uint32_t *state) : mt(state), index((int)state[624]) {}
    void store() { mt[624] = (uint32_t)index; }

    void regenerate() {
        constexpr int N = 624, M = 397;
        constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MATRIX_A = 0x9908b0dfu;
        int kk = 0;
        for (; kk < N - M; ++kk) {
            const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        for (; kk < N - 1; ++kk) {
            const uint32_t y = (mt[kk] & UPPER) | (mt[kk + 1] & LOWER);
            mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        }
        const uint32_t y = (mt[N - 1] & UPPER) | (mt[0] & LOWER);
        mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((y & 1u) ? MATRIX_A : 0u);
        index = 0;
    }

    inline uint32_t next() {
        if (index >= 624) regenerate();
        uint32_t y = mt[index++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }

    inline uint32_t randbelow(uint32_t n) {  // n >= 1
        const int bits = 32 - __builtin_clz(n);
        uint32_t r = next() >> (32 - bits);
        while (r >= n) r = next() >> (32 - bits);
        return r;
    }

    template <typename T>
    void shuffle_range(T *x, int64_t n) {
        for (int64_t i = 0; i < n; ++i) x[i] = (T)i;
        for (int64_t i = n - 1; i >= 1; --i) {
            const int64_t j = (int64_t)randbelow((uint32_t)(i + 1));
            const T t = x[i];
            x[i] = x[j];
            x[j] = t;
        }
    }
};

// CPython accepts any index in [0, 624] through random.setstate (0 right after its own regenerate)
bool valid_state(const uint32_t *s) { return s != nullptr && s[624] <= 624; }

struct SelectJob {
    const int64_t *I;
    int64_t nq;
    int k;
    const int64_t *p2id;
    int64_t n_rows;
    const int64_t *pos_pid;
    const uint8_t *active;
    int negative_sample, n_sel;
    const uint16_t *orders;  // [n_active, k] or null (select_topk)
    const int64_t *order_slot;  // per query row: slot in orders
    int64_t *out_neg;
    int32_t *out_cnt;
    uint16_t *hit_mask;  // bit r set: the positive was met at rank r + 1 (<= 10) before the walk stopped
    int bad;
};

// The reference's walk (drivers/run_ann_data_gen.py:376-392), one query row.
void select_row(SelectJob &J, int64_t row, int64_t *pids) {
    const int64_t *Ir = J.I + row * J.k;
    const uint16_t *ord = J.orders ? J.orders + J.order_slot[row] * J.k : nullptr;
    const int n_sel = J.n_sel;
    for (int j = 0; j < n_sel; ++j) {  // gather first: independent loads overlap their DRAM latency
        int64_t r = Ir[ord ? ord[j] : j];
        if (r < 0) r += J.n_rows;  // numpy negative index: faiss pads with -1, the reference then reads the last row
        if (r < 0 || r >= J.n_rows) {
            J.bad = 1;
            r = 0;
        }
        pids[j] = J.p2id[r];
    }
    const int64_t pos = J.pos_pid[row];
    int64_t *neg = J.out_neg + row * J.negative_sample;
    int cnt = 0;
    uint16_t mask = 0;
    for (int j = 0; j < n_sel; ++j) {
        const int64_t p = pids[j];
        if (p == pos) {
            if (j < 10) mask |= (uint16_t)(1u << j);
            continue;
        }
        bool dup = false;
        for (int t = 0; t < cnt; ++t) dup |= neg[t] == p;
        if (dup) continue;
        if (cnt >= J.negative_sample) break;
        neg[cnt++] = p;
    }
    for (int t = cnt; t < J.negative_sample; ++t) neg[t] = -1;
    J.out_cnt[row] = cnt;
    J.hit_mask[row] = mask;
}

}  // namespace

}  // namespace ance

using namespace ance;

extern "C" int ance_host_py_shuffle(uint32_t *mt_state, int64_t n, int64_t *out) {
    if (!valid_state(mt_state) || n < 0 || n > 0x7fffffffll || (n > 0 && out == nullptr)) {
        set_last_error("ance_host_py_shuffle: bad state or n");
        return ANCE_E_INVALID;
    }
    PyMT g(mt_state);
    g.shuffle_range(out, n);
    g.store();
    return ANCE_OK;
}

extern "C" int ance_host_select_negatives(uint32_t *mt_state, const int64_t *I, int64_t nq, int k, const int64_t *p2id,
                                          int64_t n_rows, const int64_t *pos_pid, const uint8_t *active,
                                          int negative_sample, int select_topk, int n_threads, int64_t *out_neg,
                                          int32_t *out_cnt, double *out_mrr) {
    if (nq < 0 || k < 1 || k > 65535 || negative_sample < 0 || n_rows < 1 || (!select_topk && !valid_state(mt_state)) ||
        (nq > 0 && (!I || !p2id || !pos_pid || !active || !out_cnt || (negative_sample > 0 && !out_neg)))) {
        set_last_error("ance_host_select_negatives: bad arguments");
        return ANCE_E_INVALID;
    }
    SelectJob J{};
    J.I = I; J.nq = nq; J.k = k; J.p2id = p2id; J.n_rows = n_rows; J.pos_pid = pos_pid; J.active = active;
    J.negative_sample = negative_sample;
    J.n_sel = select_topk ? std::min(k, negative_sample + 1) : k;
    J.out_neg = out_neg; J.out_cnt = out_cnt;
    std::vector<uint16_t> orders, mask((size_t)nq, 0);
    std::vector<int64_t> slot;
    J.hit_mask = mask.data();
    if (!select_topk) {
        // phase 1 (sequential, data independent): one shuffled order per effective query row
        slot.assign((size_t)nq, -1);
        int64_t n_active = 0;
        for (int64_t r = 0; r < nq; ++r)
            if (active[r]) slot[r] = n_active++;
        orders.resize((size_t)n_active * k);
        PyMT g(mt_state);
        for (int64_t s = 0; s < n_active; ++s) g.shuffle_range(orders.data() + s * k, k);
        g.store();
        J.orders = orders.data();
        J.order_slot = slot.data();
    }
    // phase 2 (parallel over query rows)
    int T = n_threads > 0 ? n_threads : (int)std::min<unsigned>(16u, std::max(1u, std::thread::hardware_concurrency()));
    if (nq < 4096) T = 1;
    std::vector<int> bad((size_t)T, 0);
    auto work = [&](int t) {
        SelectJob Jt = J;
        std::vector<int64_t> pids((size_t)J.n_sel);
        const int64_t r0 = nq * t / T, r1 = nq * (t + 1) / T;
        for (int64_t r = r0; r < r1; ++r) {
            if (!active[r]) {
                out_cnt[r] = -1;
                for (int j = 0; j < negative_sample; ++j) out_neg[r * negative_sample + j] = -1;
                continue;
            }
            select_row(Jt, r, pids.data());
        }
        bad[t] = Jt.bad;
    };
    if (T == 1) {
        work(0);
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < T; ++t) th.emplace_back(work, t);
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t)
        if (bad[t]) {
            set_last_error("ance_host_select_negatives: neighbour row id out of range");
            return ANCE_E_INVALID;
        }
    if (out_mrr) {  // the reference's accumulation order: queries in row order, ranks ascending
        double mrr = 0.0;
        for (int64_t r = 0; r < nq; ++r)
            for (int j = 0; j < 10 && mask[r]; ++j)
                if (mask[r] >> j & 1) {
                    mrr += 1.0 / (double)(j + 1);
                    if (!(mask[r] >> (j + 1))) break;
                }
        *out_mrr = mrr;
    }
    return ANCE_OK;
}

extern "C" int ance_host_write_ann_training(const char *path, const int64_t *order, int64_t n_order, const int64_t *qid,
                                            const int64_t *pos_pid, const int64_t *src_row, const int64_t *neg,
                                            const int32_t *cnt, int negative_sample, int64_t *out_lines) {
    if (!path || n_order < 0 || (n_order > 0 && (!order || !qid || !pos_pid || !src_row || !cnt)) || negative_sample < 0) {
        set_last_error("ance_host_write_ann_training: bad arguments");
        return ANCE_E_INVALID;
    }
    FILE *f = fopen(path, "w");
    if (!f) {
        set_last_error((std::string("ance_host_write_ann_training: cannot open ") + path).c_str());
        return ANCE_E_INVALID;
    }
    std::vector<char> buf(1 << 20);
    setvbuf(f, buf.data(), _IOFBF, buf.size());
    std::string line;
    char num[32];
    int64_t lines = 0;
    for (int64_t t = 0; t < n_order; ++t) {
        const int64_t row = order[t];
        const int64_t src = src_row[row];
        if (src < 0) continue;  // query not effective / without a positive
        line.clear();
        line.append(num, (size_t)snprintf(num, sizeof num, "%lld\t", (long long)qid[row]));
        line.append(num, (size_t)snprintf(num, sizeof num, "%lld\t", (long long)pos_pid[row]));
        const int c = cnt[src];
        for (int j = 0; j < c; ++j)
            line.append(num, (size_t)snprintf(num, sizeof num, j ? ",%lld" : "%lld", (long long)neg[src * negative_sample + j]));
        line.push_back('\n');
        if (fwrite(line.data(), 1, line.size(), f) != line.size()) {
            fclose(f);
            set_last_error("ance_host_write_ann_training: short write");
            return ANCE_E_INVALID;
        }
        ++lines;
    }
    if (fclose(f) != 0) {
        set_last_error("ance_host_write_ann_training: close failed");
        return ANCE_E_INVALID;
    }
    if (out_lines) *out_lines = lines;
    return ANCE_OK;
}
