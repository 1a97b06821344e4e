// Native session -> FlatBatch builder (CPU, re-entrant, no dependencies).
//
// Replaces the per-session Python dict/Counter + dgl.graph + dgl.batch work of the reference's
// collate (src/utils/data/collate.py:29-256), which becomes the end-to-end limiter once a training
// step takes a few milliseconds on the GPU.  Produces exactly the buffer the
// Python builder in sessionrec-pytorch_amd/collate.py produces (tests compare them bit for bit):
//   [32 header counts | fields, each 16-byte aligned, each with a capacity]
// Graph semantics (node ids = rank of the item among the session's distinct items, first-occurrence
// edge de-duplication, k-gram nodes, dummy nodes for sessions shorter than k, ...) follow collate.py
// line by line; see the Python file for the citations.
//
// C ABI:
//   srec_collate(kind, seqs, offs, B, order, caps[4] or NULL, out, out_cap, field_info, max_fields, n_fields)
//     kind: 0 session graph, 1 EOP multigraph, 2 shortcut graph, 3 CCS heterograph of the given order
//     seqs/offs: concatenated click sequences (int64) and B+1 offsets
//     caps: {B, N, E, U} capacities (padded layout) or NULL (exact layout)
//     out: int32 buffer of out_cap elements; returns the number of int32 written, or -needed if too small
//     field_info[f] = {offset, capacity, live_count} in the fixed field order documented below
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <map>
#include <numeric>
#include <unordered_map>
#include <utility>
#include <vector>

namespace {

using vec = std::vector<int64_t>;
constexpr int HEADER = 32;
constexpr int CHUNK = 16;

struct Field {
    vec data;
    long cap = -1;      // -1: exact
    int pad = 0;        // 0 zeros, 1 repeat last, 2 fill -1
};

inline long align4(long n) { return (n + 3) & ~3L; }

// rank of each click among the session's distinct items (np.unique), plus the sorted items
void rank_items(const int64_t* s, int L, vec& items, std::vector<int>& nid) {
    items.assign(s, s + L);
    std::sort(items.begin(), items.end());
    items.erase(std::unique(items.begin(), items.end()), items.end());
    nid.resize(L);
    for (int i = 0; i < L; ++i) nid[i] = (int)(std::lower_bound(items.begin(), items.end(), s[i]) - items.begin());
}

struct PairHash {
    size_t operator()(const std::pair<int, int>& p) const { return ((size_t)p.first << 32) ^ (size_t)(uint32_t)p.second; }
};

// distinct pairs in first-occurrence order with multiplicities (Counter(...).keys()/.values())
void dedup(const std::vector<std::pair<int, int>>& pairs, std::vector<std::pair<int, int>>& out, std::vector<int>* cnt) {
    std::unordered_map<std::pair<int, int>, int, PairHash> pos;
    out.clear();
    if (cnt) cnt->clear();
    for (auto& p : pairs) {
        auto it = pos.find(p);
        if (it == pos.end()) {
            pos.emplace(p, (int)out.size());
            out.push_back(p);
            if (cnt) cnt->push_back(1);
        } else if (cnt) {
            (*cnt)[it->second]++;
        }
    }
}

// stable grouping of edge ids by key: ptr[n+1], idx[E]
void csr(const vec& key, long n, vec& ptr, vec& idx) {
    ptr.assign(n + 1, 0);
    for (auto k : key) ptr[k + 1]++;
    for (long i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    idx.resize(key.size());
    vec cur(ptr.begin(), ptr.end() - 1);
    for (size_t e = 0; e < key.size(); ++e) idx[cur[key[e]]++] = (int64_t)e;
}

// distinct items + positions (stable), chunked
// inv[p] = slot of position p's item in `items` (-1: a padded position) - the map the row-sharded lookup gathers through
void uniq_csr(const vec& gidx, vec& items, vec& ptr, vec& pos, vec& inv, vec& cptr, vec& chunk_ptr) {
    std::vector<int64_t> live;
    for (size_t i = 0; i < gidx.size(); ++i)
        if (gidx[i] >= 0) live.push_back((int64_t)i);
    std::stable_sort(live.begin(), live.end(), [&](int64_t a, int64_t b) { return gidx[a] < gidx[b]; });
    items.clear(); ptr.assign(1, 0); pos = live;
    inv.assign(gidx.size(), -1);
    for (size_t i = 0; i < live.size(); ++i) {
        if (i == 0 || gidx[live[i]] != gidx[live[i - 1]]) {
            if (i) ptr.push_back((int64_t)i);
            items.push_back(gidx[live[i]]);
        }
        inv[live[i]] = (int64_t)items.size() - 1;
    }
    if (!live.empty()) ptr.push_back((int64_t)live.size());
    const long U = (long)items.size();
    cptr.assign(U + 1, 0);
    chunk_ptr.clear();
    for (long u = 0; u < U; ++u) {
        const int64_t cnt = ptr[u + 1] - ptr[u];
        const int64_t nch = (cnt + CHUNK - 1) / CHUNK;
        cptr[u + 1] = cptr[u] + nch;
        for (int64_t k = 0; k < nch; ++k) chunk_ptr.push_back(ptr[u] + k * CHUNK);
    }
    chunk_ptr.push_back(U ? ptr[U] : 0);
}

struct Builder {
    std::vector<std::pair<const char*, Field>> fields;   // ordered like the Python dicts
    std::vector<std::pair<const char*, int64_t>> counts;
    Field& add(const char* name, vec&& v, int pad = 0) {
        fields.emplace_back(name, Field{std::move(v), -1, pad});
        return fields.back().second;
    }
    void cap(const char* name, long c) {
        for (auto& f : fields)
            if (!strcmp(f.first, name)) f.second.cap = c;
    }
    // the distinct-item list is the LAST field of a batch: the labels the loader writes behind the batch then follow it without a
    // gap (capacities are multiples of 4 words) and (items | labels) is one contiguous request list for the row-sharded lookup
    void move_last(const char* name) {
        for (size_t i = 0; i + 1 < fields.size(); ++i)
            if (!strcmp(fields[i].first, name)) {
                std::rotate(fields.begin() + i, fields.begin() + i + 1, fields.end());
                return;
            }
    }
    long emit(int32_t* out, long out_cap, int64_t* info, int max_fields, int* n_fields) {
        long off = HEADER;
        std::vector<long> offs;
        for (auto& f : fields) {
            long cap = std::max<long>((long)f.second.data.size(), f.second.cap);
            offs.push_back(off);
            off += align4(std::max<long>(cap, 1));
        }
        *n_fields = (int)fields.size();
        if (off > out_cap || (int)fields.size() > max_fields) return -off;
        std::memset(out, 0, sizeof(int32_t) * off);
        for (size_t i = 0; i < counts.size() && i < (size_t)HEADER; ++i) out[i] = (int32_t)counts[i].second;
        for (size_t i = 0; i < fields.size(); ++i) {
            Field& f = fields[i].second;
            const long n = (long)f.data.size(), cap = std::max<long>(n, f.cap);
            int32_t* o = out + offs[i];
            for (long j = 0; j < n; ++j) o[j] = (int32_t)f.data[j];
            if (n < cap) {
                int32_t fill = 0;
                if (f.pad == 1) fill = n ? (int32_t)f.data[n - 1] : 0;
                else if (f.pad == 2) fill = -1;
                for (long j = n; j < cap; ++j) o[j] = fill;
            }
            info[3 * i + 0] = offs[i];
            info[3 * i + 1] = cap;
            info[3 * i + 2] = n;
        }
        return off;
    }
};

long build_homogeneous(int kind, const int64_t* seqs, const int64_t* offs, int B, const int64_t* caps, int32_t* out,
                       long out_cap, int64_t* info, int max_fields, int* n_fields) {
    vec seg(B + 1, 0), eseg(B + 1, 0), src, dst, iid, last, ew;
    vec items;
    std::vector<int> nid;
    std::vector<std::pair<int, int>> pairs, uniq;
    std::vector<int> cnt;
    for (int b = 0; b < B; ++b) {
        const int64_t* s = seqs + offs[b];
        const int L = (int)(offs[b + 1] - offs[b]);
        rank_items(s, L, items, nid);
        const int64_t base = seg[b];
        seg[b + 1] = base + (int64_t)items.size();
        long ne = 0;
        if (kind == 1) {                                    // EOP multigraph: every transition, click order
            for (int i = 0; i + 1 < L; ++i) { src.push_back(base + nid[i]); dst.push_back(base + nid[i + 1]); ++ne; }
        } else if (kind == 2) {                             // shortcut graph: distinct (i<=j) pairs
            pairs.clear();
            for (int i = 0; i < L; ++i)
                for (int j = i; j < L; ++j) pairs.emplace_back(nid[i], nid[j]);
            dedup(pairs, uniq, nullptr);
            for (auto& p : uniq) { src.push_back(base + p.first); dst.push_back(base + p.second); ++ne; }
        } else {                                            // session graph: distinct transitions + counts
            pairs.clear();
            for (int i = 0; i + 1 < L; ++i) pairs.emplace_back(nid[i], nid[i + 1]);
            dedup(pairs, uniq, &cnt);
            if (uniq.empty()) { uniq.emplace_back(0, 0); cnt.assign(1, 1); }
            for (size_t e = 0; e < uniq.size(); ++e) {
                src.push_back(base + uniq[e].first); dst.push_back(base + uniq[e].second); ew.push_back(cnt[e]); ++ne;
            }
        }
        eseg[b + 1] = eseg[b] + ne;
        if (kind != 2) {
            for (auto it : items) iid.push_back(it);
            last.push_back(base + nid[L - 1]);
        }
    }
    const long N = seg[B], E = (long)src.size();
    vec in_ptr, in_idx, out_ptr, out_idx;
    csr(dst, N, in_ptr, in_idx);
    csr(src, N, out_ptr, out_idx);
    Builder bd;
    bd.add("seg", vec(seg), 1); bd.add("eseg", vec(eseg), 1);
    bd.add("esrc", vec(src)); bd.add("edst", vec(dst));
    bd.add("in_ptr", std::move(in_ptr), 1); bd.add("in_idx", std::move(in_idx));
    bd.add("out_ptr", std::move(out_ptr), 1); bd.add("out_idx", std::move(out_idx));
    bd.counts = {{"B", B}, {"N", N}, {"E", E}};
    if (kind != 2) {
        vec ui, up, upos, uinv, cptr, chptr;
        uniq_csr(iid, ui, up, upos, uinv, cptr, chptr);
        bd.counts.push_back({"U", (int64_t)ui.size()});
        bd.counts.push_back({"C", (int64_t)chptr.size() - 1});
        bd.add("iid", std::move(iid), 2); bd.add("last", std::move(last), 2);
        bd.add("uniq_items", std::move(ui), 2); bd.add("uniq_ptr", std::move(up), 1); bd.add("uniq_pos", std::move(upos));
        bd.add("uniq_inv", std::move(uinv), 2);
        bd.add("uniq_cptr", std::move(cptr), 1); bd.add("chunk_ptr", std::move(chptr), 1);
    }
    if (kind == 0) bd.add("ew", std::move(ew));
    if (caps) {
        const long Bc = caps[0], Nc = caps[1], Ec = caps[2], Uc = caps[3];
        if (N > Nc || E > Ec) return 0;
        bd.cap("seg", Bc + 1); bd.cap("eseg", Bc + 1); bd.cap("esrc", Ec); bd.cap("edst", Ec);
        bd.cap("in_ptr", Nc + 1); bd.cap("in_idx", Ec); bd.cap("out_ptr", Nc + 1); bd.cap("out_idx", Ec);
        bd.cap("iid", Nc); bd.cap("last", Bc); bd.cap("uniq_items", Uc); bd.cap("uniq_ptr", Uc + 1);
        bd.cap("uniq_pos", Nc); bd.cap("uniq_inv", Nc); bd.cap("uniq_cptr", Uc + 1); bd.cap("chunk_ptr", Uc + Nc / CHUNK + 2); bd.cap("ew", Ec);
    }
    bd.move_last("uniq_items");
    return bd.emit(out, out_cap, info, max_fields, n_fields);
}

long build_ccs(const int64_t* seqs, const int64_t* offs, int B, int K, const int64_t* caps, int32_t* out, long out_cap,
               int64_t* info, int max_fields, int* n_fields) {
    // relation order = sorted((s, etype, d)) like the Python builder: for K=3:
    // (1,inter,2) (1,inter,3) (1,intra1,1) (2,inter,1) (2,intra2,2) (3,inter,1) (3,intra3,3)
    struct Rel { int s, d, kind; vec src, dst; };          // kind 0 inter, 1 intra
    std::vector<Rel> rels;
    for (int s = 1; s <= K; ++s) {
        if (s == 1) {
            for (int d = 2; d <= K; ++d) rels.push_back({1, d, 0, {}, {}});
            rels.push_back({1, 1, 1, {}, {}});
        } else {
            rels.push_back({s, 1, 0, {}, {}});
            rels.push_back({s, s, 1, {}, {}});
        }
    }
    auto find_rel = [&](int s, int d, int kind) -> Rel& {
        for (auto& r : rels)
            if (r.s == s && r.d == d && r.kind == kind) return r;
        return rels[0];
    };
    std::vector<vec> seg(K + 1, vec(B + 1, 0)), iid(K + 1), last(K + 1);
    vec items;
    std::vector<int> nid;
    std::vector<std::vector<int>> gid(K + 1);
    std::vector<std::pair<int, int>> pairs, uniq;
    for (int b = 0; b < B; ++b) {
        const int64_t* s = seqs + offs[b];
        const int L = (int)(offs[b + 1] - offs[b]);
        const int eff = std::min(K, L);
        rank_items(s, L, items, nid);
        seg[1][b + 1] = seg[1][b] + (int64_t)items.size();
        for (auto it : items) iid[1].push_back(it);
        last[1].push_back(seg[1][b] + nid[L - 1]);
        gid[1] = nid;
        for (int k = 2; k <= K; ++k) {
            std::map<std::vector<int64_t>, int> table;
            std::vector<int>& ids = gid[k];
            ids.clear();
            int ngram = 0;
            const int64_t base = seg[k][b];
            for (int j = 0; j + k <= L; ++j) {
                std::vector<int64_t> key(s + j, s + j + k);
                auto it = table.find(key);
                int g;
                if (it == table.end()) {
                    g = ngram++;
                    table.emplace(key, g);
                    if (k <= eff)
                        for (int t = 0; t < k; ++t) iid[k].push_back(s[j + t]);
                } else {
                    g = it->second;
                }
                ids.push_back(g);
            }
            if (k <= eff) {
                seg[k][b + 1] = base + ngram;
                last[k].push_back(base + ids.back());
            } else {                                       // dummy node repeating the smallest item id
                seg[k][b + 1] = base + 1;
                for (int t = 0; t < k; ++t) iid[k].push_back(items[0]);
                last[k].push_back(base);
            }
        }
        for (int k = 1; k <= eff; ++k) {                   // intra_k: gram_i -> gram_{i+1}
            pairs.clear();
            const std::vector<int>& g = gid[k];
            for (size_t i = 0; i + 1 < g.size(); ++i) pairs.emplace_back(g[i], g[i + 1]);
            dedup(pairs, uniq, nullptr);
            Rel& r = find_rel(k, k, 1);
            for (auto& p : uniq) { r.src.push_back(seg[k][b] + p.first); r.dst.push_back(seg[k][b] + p.second); }
        }
        for (int k = 2; k <= eff; ++k) {                   // inter: item_i -> gram_{i+1}, gram_i -> item_{i+k}
            const std::vector<int>& g = gid[k];
            pairs.clear();
            for (int i = 0; i < L - k; ++i) pairs.emplace_back(nid[i], g[i + 1]);
            dedup(pairs, uniq, nullptr);
            Rel& r1 = find_rel(1, k, 0);
            for (auto& p : uniq) { r1.src.push_back(seg[1][b] + p.first); r1.dst.push_back(seg[k][b] + p.second); }
            pairs.clear();
            for (int i = 0; i < L - k; ++i) pairs.emplace_back(g[i], nid[i + k]);
            dedup(pairs, uniq, nullptr);
            Rel& r2 = find_rel(k, 1, 0);
            for (auto& p : uniq) { r2.src.push_back(seg[k][b] + p.first); r2.dst.push_back(seg[1][b] + p.second); }
        }
    }
    std::vector<long> ncap(K + 1);
    for (int k = 1; k <= K; ++k) {
        ncap[k] = caps ? caps[1] : seg[k][B];
        if (seg[k][B] > ncap[k]) return 0;
    }
    Builder bd;
    bd.counts.push_back({"B", B});
    static const char* SEG[] = {"", "seg1", "seg2", "seg3", "seg4", "seg5", "seg6"};
    static const char* IID[] = {"", "iid1", "iid2", "iid3", "iid4", "iid5", "iid6"};
    static const char* LAST[] = {"", "last1", "last2", "last3", "last4", "last5", "last6"};
    static const char* LASTCAT[] = {"", "lastcat1", "lastcat2", "lastcat3", "lastcat4", "lastcat5", "lastcat6"};
    if (K > 6) return 0;
    vec gidx;
    for (int k = 1; k <= K; ++k) {
        bd.add(SEG[k], vec(seg[k]), 1);
        bd.add(IID[k], vec(iid[k]), 2);
        bd.add(LAST[k], vec(last[k]), 2);
        bd.counts.push_back({"N", seg[k][B]});
        bd.counts.push_back({"GK", seg[k][B] * k});
        vec blk((size_t)ncap[k] * k, -1);
        std::copy(iid[k].begin(), iid[k].end(), blk.begin());
        gidx.insert(gidx.end(), blk.begin(), blk.end());
    }
    vec ui, up, upos, uinv, cptr, chptr;
    uniq_csr(gidx, ui, up, upos, uinv, cptr, chptr);
    const long G = (long)gidx.size(), U = (long)ui.size(), C = (long)chptr.size() - 1;
    bd.add("gidx", std::move(gidx), 2);
    bd.add("uniq_items", std::move(ui), 2); bd.add("uniq_ptr", std::move(up), 1); bd.add("uniq_pos", std::move(upos));
    bd.add("uniq_inv", std::move(uinv), 2);
    bd.add("uniq_cptr", std::move(cptr), 1); bd.add("chunk_ptr", std::move(chptr), 1);
    bd.counts.push_back({"G", G}); bd.counts.push_back({"U", U}); bd.counts.push_back({"C", C});
    static char names[64][6][32];
    int ri = 0;
    for (auto& r : rels) {
        vec in_ptr, in_idx, out_ptr, out_idx;
        csr(r.dst, seg[r.d][B], in_ptr, in_idx);
        csr(r.src, seg[r.s][B], out_ptr, out_idx);
        bd.counts.push_back({"E", (int64_t)r.src.size()});
        bd.add("r_src", std::move(r.src)); bd.add("r_dst", std::move(r.dst));
        bd.add("r_in_ptr", std::move(in_ptr), 1); bd.add("r_in_idx", std::move(in_idx));
        bd.add("r_out_ptr", std::move(out_ptr), 1); bd.add("r_out_idx", std::move(out_idx));
        ++ri;
    }
    (void)names;
    vec offs_k(K + 1, 0);
    for (int k = 1; k <= K; ++k) offs_k[k] = offs_k[k - 1] + ncap[k];
    vec perm, cat_seg(B + 1, 0);
    for (int b = 0; b < B; ++b) {
        for (int k = 1; k <= K; ++k)
            for (int64_t i = seg[k][b]; i < seg[k][b + 1]; ++i) perm.push_back(i + offs_k[k - 1]);
        cat_seg[b + 1] = (int64_t)perm.size();
    }
    vec inv((size_t)offs_k[K], -1);
    for (size_t i = 0; i < perm.size(); ++i) inv[perm[i]] = (int64_t)i;
    const long NT = (long)perm.size();
    bd.add("cat_perm", std::move(perm), 2); bd.add("cat_inv", std::move(inv)); bd.add("cat_seg", std::move(cat_seg), 1);
    for (int k = 1; k <= K; ++k) {
        vec lc(last[k]);
        for (auto& x : lc) x += offs_k[k - 1];
        bd.add(LASTCAT[k], std::move(lc), 2);
    }
    bd.counts.push_back({"NT", NT});
    if (caps) {
        const long Bc = caps[0], N = caps[1], E = caps[2], U2 = caps[3];
        bd.cap("uniq_items", U2); bd.cap("uniq_ptr", U2 + 1); bd.cap("uniq_pos", G); bd.cap("uniq_inv", G); bd.cap("uniq_cptr", U2 + 1);
        bd.cap("chunk_ptr", U2 + G / CHUNK + 2); bd.cap("cat_perm", N * K); bd.cap("cat_seg", Bc + 1);
        for (int k = 1; k <= K; ++k) {
            bd.cap(SEG[k], Bc + 1); bd.cap(IID[k], N * k); bd.cap(LAST[k], Bc); bd.cap(LASTCAT[k], Bc);
        }
        for (auto& f : bd.fields) {
            if (!strcmp(f.first, "r_src") || !strcmp(f.first, "r_dst") || !strcmp(f.first, "r_in_idx") ||
                !strcmp(f.first, "r_out_idx")) {
                if ((long)f.second.data.size() > E) return 0;
                f.second.cap = E;
            }
            if (!strcmp(f.first, "r_in_ptr") || !strcmp(f.first, "r_out_ptr")) f.second.cap = N + 1;
        }
    }
    bd.move_last("uniq_items");
    return bd.emit(out, out_cap, info, max_fields, n_fields);
}

}  // namespace

extern "C" long srec_collate(int kind, const int64_t* seqs, const int64_t* offs, int B, int order, const int64_t* caps,
                             int32_t* out, long out_cap, int64_t* field_info, int max_fields, int* n_fields) {
    if (B <= 0) return 0;
    for (int b = 0; b < B; ++b)
        if (offs[b + 1] <= offs[b]) return 0;               // empty session: caller error
    if (kind >= 0 && kind <= 2)
        return build_homogeneous(kind, seqs, offs, B, caps, out, out_cap, field_info, max_fields, n_fields);
    if (kind == 3) return build_ccs(seqs, offs, B, order, caps, out, out_cap, field_info, max_fields, n_fields);
    return 0;
}
