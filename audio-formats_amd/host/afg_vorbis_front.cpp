// afg_vorbis_front.cpp -- host front-end for Ogg Vorbis I files.
//
// Everything the reference does ahead of the transform seam (stb_vorbis2.d:2526): Ogg page / lacing walk
// (:984-1152), identification, comment and setup headers (:2669-3266: codebooks :2791-3000, floors :3016-3090,
// residues :3092-3146, mappings :3148-3205, modes :3207-3218), and per audio packet the mode and window decision
// (:2300-2352), floor 1 decode (:2373-2462), residue decode for types 0/1/2 (:1565-1713), inverse coupling
// (:2493-2514) and the floor curve multiplication (:2255-2284, :1534-1563); then the pull API's bookkeeping: the
// first frame is only primed (:2659-2667), the last page's granule position truncates the final frame
// (:2564-2590), the stream length comes from the last page (:3797-3868).
//
// Organisation (not the reference's): the file is first split into packets (one byte range each, with the page
// facts a packet's decode depends on), then every packet is decoded from a flat bit reader; code books are
// decoded through a 10-bit table plus a binary tree for longer words, indexed by symbol (no sparse / sorted
// distinction).  Every float is produced by the reference's expression trees.  Floor 0 is not supported (the
// reference rejects it too, :3036).
#include "afg_vorbis_front.h"

#include "vorbis_front_tables.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace afg_vorbis {
namespace {

inline float bits_f32(unsigned b)
{
    float f;
    std::memcpy(&f, &b, 4);
    return f;
}

int ilog(int32_t n)                                        // :634-650: 0 for n <= 0, else floor(log2 n) + 1
{
    if (n <= 0) return 0;
    int r = 0;
    while (n) { r++; n >>= 1; }
    return r;
}

// ---- Ogg: pages -> packets ---------------------------------------------------------------------------
struct Packet {
    size_t first_piece = 0, n_pieces = 0;     // byte ranges in File-level `pieces`
    bool ends_page_run = false;               // the last packet that completes on its page, and that page has a granule
    uint32_t granule_lo = 0;
    bool on_last_page = false;                // PAGEFLAG_last_page of the page the packet ends on
    bool complete = false;
};
struct Piece { size_t off, len; };            // off == kZeros: `len` bytes the lacing promises but the file does not have
constexpr size_t kZeros = ~(size_t)0;

struct Demux {
    std::vector<Piece> pieces;
    std::vector<Packet> packets;
    size_t first_audio_page = 0;              // byte offset of the page after the headers (0 if they end mid-page)
};

uint32_t rd32(const uint8_t *p) { return p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

// Walks pages from `pos`; stops at the first thing the reference's state machine would fail on (missing capture
// pattern, bad version, truncated page, a fresh packet on a page flagged "continued").
void demux(const uint8_t *d, size_t n, Demux &dm)
{
    size_t pos = 0, n_page = 0;
    bool open_packet = false;
    Packet cur;
    while (pos + 27 <= n) {
        if (std::memcmp(d + pos, "OggS", 4) || d[pos + 4] != 0) break;
        const uint8_t flags = d[pos + 5];
        const uint32_t lo = rd32(d + pos + 6), hi = rd32(d + pos + 10);
        const int nseg = d[pos + 26];
        if (pos + 27 + (size_t)nseg > n) break;
        const uint8_t *lac = d + pos + 27;
        int last_complete = -1;
        if (lo != 0xffffffffu || hi != 0xffffffffu)
            for (int i = nseg - 1; i >= 0; --i)
                if (lac[i] < 255) { last_complete = i; break; }
        // (the page the comment header starts on is opened by start_decoder itself, stb_vorbis2.d:2732, not by start_packet :1056-1069: nobody
        //  looks at its "continued" flag)
        const bool unchecked = n_page == 1 && !open_packet;
        n_page++;
        if (!unchecked && open_packet != ((flags & 1) != 0)) {
            // continued flag without an open packet, or an open packet on a page that does not continue it
            if (open_packet) { cur.complete = false; dm.packets.push_back(cur); }
            break;
        }
        size_t body = pos + 27 + (size_t)nseg;
        bool truncated = false;
        for (int i = 0; i < nseg; i++) {
            if (!open_packet) {
                if (truncated) break;                       // the reference notices eof before it starts another packet
                cur = Packet();
                cur.first_piece = dm.pieces.size();
                open_packet = true;
            }
            size_t len = lac[i], real = len;
            if (body + len > n) { real = body < n ? n - body : 0; truncated = true; }
            if (real) {
                if (cur.n_pieces && dm.pieces.back().off != kZeros && dm.pieces.back().off + dm.pieces.back().len == body)
                    dm.pieces.back().len += real;
                else { dm.pieces.push_back(Piece{ body, real }); cur.n_pieces++; }
            }
            if (len > real) {                               // bytes past the end of the data read as zeros (get8 at eof)
                dm.pieces.push_back(Piece{ kZeros, len - real });
                cur.n_pieces++;
            }
            body += real;
            if (lac[i] < 255) {
                cur.complete = true;
                cur.ends_page_run = (i == last_complete);
                cur.granule_lo = lo;
                cur.on_last_page = (flags & 4) != 0;
                dm.packets.push_back(cur);
                open_packet = false;
            }
        }
        if (truncated) {
            if (open_packet) { cur.complete = false; dm.packets.push_back(cur); }
            open_packet = false;
            break;
        }
        pos = body;
        if (dm.packets.size() == 3 && !open_packet && !dm.first_audio_page) dm.first_audio_page = pos;
    }
}

// ---- bit reader over one packet (LSB first), with the reference's end-of-packet rules -------------------
struct Bits {
    const uint8_t *d;
    const Piece *pc;
    size_t n_pieces, piece = 0, at = 0;
    uint64_t acc = 0;
    int have = 0;
    bool dry = false;        // no more bytes
    bool invalid = false;    // a fixed-width read ran past the end (valid_bits == INVALID_BITS)
    // The reference fetches the packet a byte at a time (:1126-1184) and its accumulator is zero above the bits fetched:
    // `sv` is ITS count of fetched, unread bits at this point of the packet.  Only a word that is not in the book looks
    // at it (miss() below: the search over the sorted words compares all 32 bits of the accumulator).
    int sv = 0;
    void sv_take(int n)                                  // (called after `have -= n`: sv <= have, equal modulo 8)
    {
        if (sv < n) sv += 8 * ((n - sv + 7) / 8);
        sv -= n;
        if (sv > have) sv = have;
    }
    void sv_prefetch()                                   // prep_huffman, :1186-1198
    {
        fill();
        while (sv <= 24 && sv + 8 <= have) sv += 8;
    }
    Bits(const uint8_t *data, const Piece *p, size_t np) : d(data), pc(p), n_pieces(np) {}
    void fill()
    {
        while (have <= 56 && !dry) {
            while (piece < n_pieces && at >= pc[piece].len) { piece++; at = 0; }
            if (piece >= n_pieces) { dry = true; break; }
            acc |= (uint64_t)(pc[piece].off == kZeros ? 0 : d[pc[piece].off + at]) << have;
            at++;
            have += 8;
        }
    }
    uint32_t get(int n)                                  // get_bits, :1154-1184
    {
        if (invalid) return 0;
        if (n == 0) return 0;
        if (have < n) fill();
        if (have < n) {
            invalid = true;
            have = 0;
            acc = 0;
            sv = 0;
            return 0;
        }
        const uint32_t v = (uint32_t)(acc & ((n >= 32) ? 0xffffffffull : ((1ull << n) - 1)));
        acc >>= n;
        have -= n;
        sv_take(n);
        return v;
    }
    bool exhausted() { fill(); return have == 0; }
};

// ---- code books ----------------------------------------------------------------------------------------
struct Book {
    int dim = 0, entries = 0;
    std::vector<uint8_t> len;                 // 255 = unused
    int lookup = 0;
    bool sequence = false;
    float minimum = 0;
    std::vector<float> mult;                  // entries * dim (lookup 1, expanded) or entries * dim (lookup 2)
    // decoding: table over the low 10 bits, then a tree
    std::vector<int32_t> fast;                // >= 0: symbol; -1: none; <= -2: -(node + 2)
    struct Node { int32_t child[2]; };
    std::vector<Node> tree;                   // child: >= 0 node index, < 0: -(symbol + 1); INT32_MIN-ish 0x7fffffff: none
    bool usable = false;
    // A book whose length list is sparse and stays sparse (fewer than entries/4 words in use, :2838) is handled by the
    // reference through its *sorted* word list: vector look-ups are indexed by a word's rank in that list -- the
    // left-to-right position of its leaf -- not by its entry number.  For lookup type 1 the expansion is made in that
    // order too (consistent); for lookup type 2 the table is still laid out by entry, so such a book reads the
    // vector of a different entry (:1344-1444 with :2987-2997).  The same happens here.
    std::vector<int32_t> rank;                // empty: index by symbol
    int vec(int sym) const { return rank.empty() ? sym : rank[(size_t)sym]; }
    // The reference's sorted word list (:776-828): every word of a book that stays sparse, the words of more than 10
    // bits of any other; left-aligned, ascending.  A stream word that is NOT in the book (possible only in a book whose
    // lengths leave the tree incomplete) is read by the reference as the nearest listed word below it (:1211-1240).
    std::vector<uint32_t> sorted_word;
    std::vector<int32_t> sorted_sym;
    bool search = false;                      // the reference decodes a table miss by that search (else: exact match only)
};
constexpr int kFast = 10;
constexpr int32_t kNone = 0x7fffffff;

// Huffman words of the Vorbis I specification (section 3.2.1): symbols, in order, take the lowest-valued free
// word of their length, i.e. the leftmost free place at that depth of the binary tree (first transmitted bit =
// first branch, 0 = left).  The tree built by that insertion is also the decoder; words of up to 10 bits are
// additionally spread over a table indexed by the next 10 stream bits.
struct Builder {
    std::vector<Book::Node> &t;
    std::vector<char> full;                    // subtree below this node has no free place left
    explicit Builder(std::vector<Book::Node> &tree) : t(tree) {}
    int32_t fresh()
    {
        t.push_back(Book::Node{ { kNone, kNone } });
        full.push_back(0);
        return (int32_t)t.size() - 1;
    }
    bool closed(int32_t c) const { return c != kNone && (c < 0 || full[(size_t)c]); }
    // place `sym` `depth` levels below `node`
    bool place(int32_t node, int depth, int sym)
    {
        if (full[(size_t)node]) return false;
        bool ok = false;
        for (int v = 0; v < 2 && !ok; v++) {
            int32_t c = t[(size_t)node].child[v];
            if (depth == 1) {
                if (c != kNone) continue;
                t[(size_t)node].child[v] = -(sym + 1);
                ok = true;
                break;
            }
            if (c != kNone && c < 0) continue;             // a leaf sits here
            if (c == kNone) {
                c = fresh();
                t[(size_t)node].child[v] = c;
            }
            ok = place(c, depth - 1, sym);
        }
        full[(size_t)node] = closed(t[(size_t)node].child[0]) && closed(t[(size_t)node].child[1]);
        return ok;
    }
};

void leaf_order(const Book &b, int32_t node, std::vector<int32_t> &order)
{
    for (int v = 0; v < 2; v++) {
        const int32_t c = b.tree[(size_t)node].child[v];
        if (c == kNone) continue;
        if (c < 0) order.push_back(-(c + 1));
        else leaf_order(b, c, order);
    }
}

void leaf_words(const Book &b, int32_t node, uint32_t prefix, int depth, bool all, std::vector<uint32_t> &words,
                std::vector<int32_t> &syms)
{
    for (int v = 0; v < 2; v++) {
        const int32_t c = b.tree[(size_t)node].child[v];
        if (c == kNone) continue;
        const uint32_t p = prefix | ((uint32_t)v << (31 - depth));
        if (c >= 0) { leaf_words(b, c, p, depth + 1, all, words, syms); continue; }
        const int sym = -(c + 1);
        if (!all && b.len[(size_t)sym] <= kFast) continue;
        words.push_back(p);
        syms.push_back(sym);
    }
}

bool build_decoder(Book &b)
{
    b.fast.assign(1 << kFast, -1);
    b.tree.clear();
    Builder bu(b.tree);
    const int32_t root = bu.fresh();
    int used = 0;
    for (int s = 0; s < b.entries; s++) {
        const int l = b.len[(size_t)s];
        if (l == 255) continue;
        if (!bu.place(root, l, s)) return false;             // over-subscribed
        used++;
    }
    // table over the next 10 bits: a symbol (word <= 10 bits) or the tree node reached after 10 branches.  One walk of the
    // tree's top ten levels: whatever ends a path of d branches (a leaf, a missing child) owns every key whose low d bits are
    // that path -- 2^(10-d) entries, 2^d apart -- and a node at depth 10 owns the one key that spells its path.
    {
        struct Item { int32_t node; uint32_t path; int depth; };
        Item stack[2 * kFast + 2];
        int top = 0;
        stack[top++] = Item{ root, 0u, 0 };
        while (top) {
            const Item it = stack[--top];
            if (it.depth == kFast) { b.fast[it.path] = -(it.node + 2); continue; }
            for (int bit = 1; bit >= 0; bit--) {
                const int32_t c = b.tree[(size_t)it.node].child[bit];
                const uint32_t path = it.path | ((uint32_t)bit << it.depth);
                if (c >= 0 && c != kNone) { stack[top++] = Item{ c, path, it.depth + 1 }; continue; }
                const int32_t out = c == kNone ? -1 : -(c + 1);
                for (uint32_t key = path; key < (1u << kFast); key += 2u << it.depth) b.fast[key] = out;
            }
        }
    }
    b.usable = used > 0;
    return true;
}

// valid_bits = 0: the reference drops the bits it has fetched -- the bytes behind them are still to come (a word that is
// not in the book can fail in the middle of a packet, and the floor decode reads on, :3132-3150)
inline int no_symbol(Bits &br)
{
    br.acc = br.sv >= 64 ? 0 : br.acc >> br.sv;
    br.have -= br.sv;
    br.sv = 0;
    return -1;
}

// codebook_decode_scalar_raw's search (:1211-1240): the last listed word that is <= the 32 accumulator bits, taken
// with ITS length whether or not it is a prefix of them
int search_symbol(Bits &br, const Book &b)
{
    uint32_t a = (uint32_t)br.acc;
    if (br.sv < 32) a &= (1u << br.sv) - 1;
    uint32_t code = 0;
    for (int i = 0; i < 32; i++) code |= ((a >> i) & 1u) << (31 - i);
    int x = 0, n = (int)b.sorted_word.size();
    while (n > 1) {
        const int m = x + (n >> 1);
        if (b.sorted_word[(size_t)m] <= code) { x = m; n -= n >> 1; }
        else n >>= 1;
    }
    const int sym = b.sorted_sym[(size_t)x];
    const int l = b.len[(size_t)sym];
    if (br.sv < l) return no_symbol(br);
    br.acc >>= l;
    br.have -= l;
    br.sv -= l;
    return sym;
}

// one symbol, or -1 at the end of the packet / on a word that is not in the book (:1211-1286)
int symbol(Bits &br, const Book &b)
{
    if (!b.usable) return -1;
    if (br.have < 32) br.fill();
    if (br.sv < kFast) br.sv_prefetch();
    int32_t e = b.fast[(size_t)(br.acc & ((1u << kFast) - 1))];
    if (e >= 0) {
        const int l = b.len[(size_t)e];
        if (br.sv < l) return no_symbol(br);
        br.acc >>= l;
        br.have -= l;
        br.sv -= l;
        return e;
    }
    br.sv_prefetch();
    if (b.search) return search_symbol(br, b);
    if (e == -1) return no_symbol(br);
    int32_t node = -(e + 2);
    int used = kFast;
    for (;;) {
        if (used >= br.sv) return no_symbol(br);
        const int v = (int)((br.acc >> used) & 1);
        used++;
        const int32_t nx = b.tree[(size_t)node].child[v];
        if (nx == kNone) return no_symbol(br);
        if (nx < 0) {
            br.acc >>= used;
            br.have -= used;
            br.sv -= used;
            return -(nx + 1);
        }
        node = nx;
    }
}

float unpack_float(uint32_t x)                             // float32_unpack, :662-671
{
    const uint32_t mant = x & 0x1fffff, sign = x & 0x80000000u, e = (x & 0x7fe00000u) >> 21;
    const double res = sign ? -(double)mant : (double)mant;
    return (float)std::ldexp((float)res, (int)e - 788);
}

int lookup1_values(int entries, int dim)                   // :838-848
{
    int r = (int)std::floor(std::exp((float)std::log((float)entries) / dim));
    if ((int)std::floor(std::pow((float)r + 1, dim)) <= entries) ++r;
    if (std::pow((float)r + 1, dim) <= entries) return -1;
    if ((int)std::floor(std::pow((float)r, dim)) > entries) return -1;
    return r;
}

struct Floor1 {
    int partitions = 0, multiplier = 1, rangebits = 0, values = 0;
    uint8_t part_class[32];
    uint8_t class_dim[16], class_sub[16], class_master[16];
    int16_t sub_books[16][8];
    uint16_t x[256];
    uint8_t order[256], lo[256], hi[256];
};
struct Residue {
    uint32_t begin = 0, end = 0, part_size = 1;
    int classifications = 1, classbook = 0, type = 0;
    int max_dim = 1;                         // largest dimension among its value books
    int16_t books[64][8];
};
struct Mapping {
    int coupling = 0, submaps = 1;
    uint32_t step_off = 0;                   // where this mapping's coupling steps sit in File::fl_steps
    uint8_t mag[256], ang[256], mux[16];
    uint8_t floor_of[16], residue_of[16];
};
struct Mode { int blockflag = 0, mapping = 0; };

struct Setup {
    int channels = 0, bs[2] = { 0, 0 };
    unsigned rate = 0;
    std::vector<Book> books;
    std::vector<Floor1> floors;
    std::vector<Residue> residues;
    std::vector<Mapping> mappings;
    std::vector<Mode> modes;
};

bool read_setup(Bits &br, Setup &st)
{
    const int nbooks = (int)br.get(8) + 1;
    st.books.resize((size_t)nbooks);
    for (Book &b : st.books) {
        if (br.get(8) != 0x42 || br.get(8) != 0x43 || br.get(8) != 0x56) return false;
        uint32_t lo8 = br.get(8);
        b.dim = (int)((br.get(8) << 8) + lo8);
        lo8 = br.get(8);
        const uint32_t mid = br.get(8);
        b.entries = (int)((br.get(8) << 16) + (mid << 8) + lo8);
        const bool ordered = br.get(1) != 0;
        const bool sparse = ordered ? false : br.get(1) != 0;
        if (b.dim == 0 && b.entries != 0) return false;
        b.len.assign((size_t)b.entries, 255);
        if (ordered) {
            int cur = 0, l = (int)br.get(5) + 1;
            while (cur < b.entries) {
                const int cnt = (int)br.get(ilog(b.entries - cur));
                if (l >= 32 || cur + cnt > b.entries) return false;
                std::fill(b.len.begin() + cur, b.len.begin() + cur + cnt, (uint8_t)l);
                cur += cnt;
                ++l;
            }
        } else {
            for (int j = 0; j < b.entries; j++) {
                if (!sparse || br.get(1)) {
                    b.len[(size_t)j] = (uint8_t)(br.get(5) + 1);
                    if (b.len[(size_t)j] == 32) return false;
                }
            }
        }
        if (!build_decoder(b)) return false;
        int in_use = 0;
        for (uint8_t l : b.len) in_use += l != 255;
        const bool stays_sparse = sparse && in_use < (b.entries >> 2);
        std::vector<int32_t> order;                          // symbols by leaf position
        if (stays_sparse) {
            leaf_order(b, 0, order);
            b.rank.assign((size_t)b.entries, 0);
            for (size_t k = 0; k < order.size(); k++) b.rank[(size_t)order[k]] = (int32_t)k;
        }
        leaf_words(b, 0, 0u, 0, stays_sparse, b.sorted_word, b.sorted_sym);
        b.search = b.entries > 8 ? !b.sorted_word.empty() : stays_sparse;     // :1214
        b.lookup = (int)br.get(4);
        if (b.lookup > 2) return false;
        if (b.lookup > 0) {
            b.minimum = unpack_float(br.get(32));
            const float delta = unpack_float(br.get(32));
            const int value_bits = (int)br.get(4) + 1;
            b.sequence = br.get(1) != 0;
            uint32_t nvals;
            if (b.lookup == 1) {
                const int v = lookup1_values(b.entries, b.dim);
                if (v < 0) return false;
                nvals = (uint32_t)v;
            } else {
                nvals = (uint32_t)b.entries * (uint32_t)b.dim;
            }
            if (nvals == 0) return false;
            // (the reference decodes inside a fixed 220 KB arena, stream.d:1683: set-ups anywhere near these caps fail
            //  there long before; the caps only keep damaged headers from asking for terabytes)
            if (nvals > (1u << 24) || (uint64_t)b.entries * (uint64_t)b.dim > (1u << 24)) return false;
            std::vector<uint16_t> raw(nvals);
            for (uint32_t j = 0; j < nvals; j++) {
                raw[j] = (uint16_t)br.get(value_bits);
                if (br.invalid) return false;
            }
            b.mult.assign((size_t)b.entries * (size_t)b.dim, 0.0f);
            float last = 0;                                  // carried across entries when sequence_p is set: :2974, :2994
            if (b.lookup == 1) {
                const int count = stays_sparse ? (int)order.size() : b.entries;
                for (int jj = 0; jj < count; jj++) {
                    const int j = stays_sparse ? order[(size_t)jj] : jj;      // entry whose vector goes to slot jj
                    uint32_t div = 1;
                    for (int k = 0; k < b.dim; k++) {
                        const uint32_t off = ((uint32_t)j / div) % nvals;
                        const float val = raw[off] * delta + b.minimum + last;
                        b.mult[(size_t)jj * (size_t)b.dim + (size_t)k] = val;
                        if (b.sequence) last = val;
                        if (k + 1 < b.dim) {
                            if (div > 0xffffffffu / nvals) return false;
                            div *= nvals;
                        }
                    }
                }
                b.lookup = 2;
            } else {
                for (uint32_t j = 0; j < nvals; j++) {
                    const float val = raw[j] * delta + b.minimum + last;
                    b.mult[j] = val;
                    if (b.sequence) last = val;
                }
            }
        }
    }
    const int ntime = (int)br.get(6) + 1;
    for (int i = 0; i < ntime; i++)
        if (br.get(16) != 0) return false;

    const int nfloors = (int)br.get(6) + 1;
    st.floors.resize((size_t)nfloors);
    for (Floor1 &g : st.floors) {
        const uint32_t type = br.get(16);
        if (type != 1) return false;                         // type 0: unsupported by the reference, > 1 invalid
        g.partitions = (int)br.get(5);
        int max_class = -1;
        for (int j = 0; j < g.partitions; j++) {
            g.part_class[j] = (uint8_t)br.get(4);
            max_class = std::max(max_class, (int)g.part_class[j]);
        }
        for (int j = 0; j <= max_class; j++) {
            g.class_dim[j] = (uint8_t)(br.get(3) + 1);
            g.class_sub[j] = (uint8_t)br.get(2);
            g.class_master[j] = 0;
            if (g.class_sub[j]) {
                g.class_master[j] = (uint8_t)br.get(8);
                if (g.class_master[j] >= nbooks) return false;
            }
            for (int k = 0; k < 1 << g.class_sub[j]; k++) {
                g.sub_books[j][k] = (int16_t)((int)br.get(8) - 1);
                if (g.sub_books[j][k] >= nbooks) return false;
            }
        }
        g.multiplier = (int)br.get(2) + 1;
        g.rangebits = (int)br.get(4);
        g.x[0] = 0;
        g.x[1] = (uint16_t)(1 << g.rangebits);
        g.values = 2;
        for (int j = 0; j < g.partitions; j++)
            for (int k = 0; k < g.class_dim[g.part_class[j]]; k++) {
                if (g.values >= 250) return false;
                g.x[g.values++] = (uint16_t)br.get(g.rangebits);
            }
        // order of the points along x, duplicates are invalid
        for (int j = 0; j < g.values; j++) g.order[j] = (uint8_t)j;
        std::stable_sort(g.order, g.order + g.values, [&](uint8_t a, uint8_t c) { return g.x[a] < g.x[c]; });
        for (int j = 0; j + 1 < g.values; j++)
            if (g.x[g.order[j]] == g.x[g.order[j + 1]]) return false;
        for (int j = 2; j < g.values; j++) {                 // nearest lower / higher neighbour among the earlier points
            int lo = 0, hi = 0, lov = -1, hiv = 65536;
            for (int i = 0; i < j; i++) {
                if (g.x[i] > lov && g.x[i] < g.x[j]) { lo = i; lov = g.x[i]; }
                if (g.x[i] < hiv && g.x[i] > g.x[j]) { hi = i; hiv = g.x[i]; }
            }
            g.lo[j] = (uint8_t)lo;
            g.hi[j] = (uint8_t)hi;
        }
    }

    const int nres = (int)br.get(6) + 1;
    st.residues.resize((size_t)nres);
    for (Residue &r : st.residues) {
        r.type = (int)br.get(16);
        if (r.type > 2) return false;
        r.begin = br.get(24);
        r.end = br.get(24);
        if (r.end < r.begin) return false;
        r.part_size = br.get(24) + 1;
        r.classifications = (int)br.get(6) + 1;
        r.classbook = (int)br.get(8);
        if (r.classbook >= nbooks) return false;
        uint8_t cascade[64];
        for (int j = 0; j < r.classifications; j++) {
            const uint32_t low = br.get(3);
            const uint32_t high = br.get(1) ? br.get(5) : 0;
            cascade[j] = (uint8_t)(high * 8 + low);
        }
        for (int j = 0; j < r.classifications; j++)
            for (int k = 0; k < 8; k++) {
                r.books[j][k] = -1;
                if (cascade[j] & (1 << k)) {
                    r.books[j][k] = (int16_t)br.get(8);
                    if (r.books[j][k] >= nbooks) return false;
                    r.max_dim = std::max(r.max_dim, st.books[(size_t)r.books[j][k]].dim);
                }
            }
    }

    const int nmap = (int)br.get(6) + 1;
    st.mappings.resize((size_t)nmap);
    for (Mapping &m : st.mappings) {
        if (br.get(16) != 0) return false;
        m.submaps = br.get(1) ? (int)br.get(4) + 1 : 1;
        m.coupling = 0;
        if (br.get(1)) {
            m.coupling = (int)br.get(8) + 1;
            if (m.coupling > st.channels) return false;
            for (int k = 0; k < m.coupling; k++) {
                m.mag[k] = (uint8_t)br.get(ilog(st.channels - 1));
                m.ang[k] = (uint8_t)br.get(ilog(st.channels - 1));
                if (m.mag[k] >= st.channels || m.ang[k] >= st.channels || m.mag[k] == m.ang[k]) return false;
            }
        }
        if (br.get(2)) return false;
        for (int j = 0; j < st.channels; j++) m.mux[j] = 0;
        if (m.submaps > 1)
            for (int j = 0; j < st.channels; j++) {
                m.mux[j] = (uint8_t)br.get(4);
                if (m.mux[j] >= m.submaps) return false;
            }
        for (int j = 0; j < m.submaps; j++) {
            br.get(8);
            m.floor_of[j] = (uint8_t)br.get(8);
            m.residue_of[j] = (uint8_t)br.get(8);
            if (m.floor_of[j] >= nfloors || m.residue_of[j] >= nres) return false;
        }
    }
    const int nmodes = (int)br.get(6) + 1;
    st.modes.resize((size_t)nmodes);
    for (Mode &m : st.modes) {
        m.blockflag = (int)br.get(1);
        const uint32_t wt = br.get(16), tt = br.get(16);
        m.mapping = (int)br.get(8);
        if (wt != 0 || tt != 0 || m.mapping >= nmap) return false;
    }
    return true;                                             // (a setup packet that ends early is read as zeros, like the reference)
}

// ---- one audio packet -------------------------------------------------------------------------------------
void render_line(float *out, int x0, int y0, int x1, int y1, int n)        // draw_line, :1534-1563
{
    const int dy = y1 - y0, adx = x1 - x0;
    int ady = std::abs(dy), x = x0, y = y0, err = 0;
    const int base = dy / adx, sy = dy < 0 ? base - 1 : base + 1;
    ady -= std::abs(base) * adx;
    if (x1 > n) x1 = n;
    if (x < x1) {
        out[x] *= bits_f32(k_inverse_db_bits[y & 255]);
        for (++x; x < x1; ++x) {
            err += ady;
            if (err >= adx) { err -= adx; y += sy; }
            else y += base;
            out[x] *= bits_f32(k_inverse_db_bits[y & 255]);
        }
    }
}

// vector of one code word added into `out` (codebook_decode, :1344-1370): note the sequence rule of this form
bool add_vector(Bits &br, const Book &b, float *out, int len)
{
    if (b.lookup == 0) return false;
    const int z = symbol(br, b);
    if (z < 0) return false;
    len = std::min(len, b.dim);
    const float *m = b.mult.data() + (size_t)b.vec(z) * (size_t)b.dim;
    if (b.sequence) {
        float last = 0;
        for (int i = 0; i < len; i++) {
            const float val = m[i] + last;
            out[i] += val;
            last = val + b.minimum;
        }
    } else {
        for (int i = 0; i < len; i++) out[i] += m[i] + 0.0f;
    }
    return true;
}
bool add_vector_strided(Bits &br, const Book &b, float *out, int len, int step)   // codebook_decode_step, :1372-1387
{
    if (b.lookup == 0) return false;
    const int z = symbol(br, b);
    if (z < 0) return false;
    len = std::min(len, b.dim);
    const float *m = b.mult.data() + (size_t)b.vec(z) * (size_t)b.dim;
    float last = 0;
    for (int i = 0; i < len; i++) {
        const float val = m[i] + last;
        out[i * step] += val;
        if (b.sequence) last = val;
    }
    return true;
}

struct Scratch {
    std::vector<float> spec;                 // channels * n/2
    std::vector<int16_t> y;                  // channels * 256
    std::vector<int> cls;                    // residue classifications of the submap being decoded
    // device floor: what the packet leaves to afg_vorbis_floor_hip
    bool device_floor = false;
    int mapping = 0;
    std::vector<uint32_t> n_points;          // per channel; 0 = really_zero_channel
    std::vector<int32_t> points;             // (x, y) pairs, channel after channel
};

// returns false when the packet cannot be decoded at all (not an audio packet / bad mode); `flags` and the window
// bounds come back for the bookkeeping
bool decode_packet(const uint8_t *d, const Demux &dm, const Packet &pk, const Setup &st, Scratch &sc, unsigned &flags, int &n_out,
                   bool &audio)
{
    Bits br(d, dm.pieces.data() + pk.first_piece, pk.n_pieces);
    audio = true;
    if (br.get(1) != 0) { audio = false; return true; }       // not audio: skipped (:2312-2315)
    const int mi = (int)br.get(ilog((int)st.modes.size() - 1));
    if (mi >= (int)st.modes.size()) return false;
    const Mode &mode = st.modes[(size_t)mi];
    int prev = 0, next = 0;
    if (mode.blockflag) {
        prev = (int)br.get(1);
        next = (int)br.get(1);
    }
    flags = (mode.blockflag ? AFG_VORBIS_LONG : 0u) | (prev ? AFG_VORBIS_PREV : 0u) | (next ? AFG_VORBIS_NEXT : 0u);
    const int n = st.bs[mode.blockflag], n2 = n >> 1, C = st.channels;
    n_out = n;
    const Mapping &map = st.mappings[(size_t)mode.mapping];
    // a channel owns n floats although its spectrum is n/2: a type-2 residue on a single channel is bounded by 2 * n/2
    // (:1594) and may write into the upper half, as it does in the reference's blocksize-sized channel buffers
    // Only the n/2 values of a channel's spectrum are ever read; the upper half exists for that one residue shape, which
    // needs it defined (it accumulates): the whole slot is cleared only when the mapping has such a submap.
    if (sc.spec.size() < (size_t)C * (size_t)n) sc.spec.resize((size_t)C * (size_t)n);
    {
        bool upper = false;
        int per_submap[16] = { 0 };
        for (int j = 0; j < C; j++) per_submap[map.mux[j]]++;
        for (int sm = 0; sm < map.submaps; sm++)
            upper = upper || (per_submap[sm] == 1 && st.residues[map.residue_of[sm]].type == 2);
        const size_t clear = upper ? (size_t)n : (size_t)n2;
        for (int j = 0; j < C; j++) std::memset(sc.spec.data() + (size_t)j * (size_t)n, 0, clear * sizeof(float));
    }
    sc.y.assign((size_t)C * 256, 0);
    bool zero[256], really_zero[256];

    // floors (:2373-2462)
    for (int i = 0; i < C; i++) {
        const Floor1 &g = st.floors[map.floor_of[map.mux[i]]];
        zero[i] = false;
        if (!br.get(1)) { zero[i] = true; continue; }
        static const int ranges[4] = { 256, 128, 86, 64 };
        const int range = ranges[g.multiplier - 1];
        int16_t *Y = sc.y.data() + (size_t)i * 256;
        int off = 2;
        Y[0] = (int16_t)br.get(ilog(range) - 1);
        Y[1] = (int16_t)br.get(ilog(range) - 1);
        for (int j = 0; j < g.partitions; j++) {
            const int pc = g.part_class[j], cdim = g.class_dim[pc], cbits = g.class_sub[pc], csub = (1 << cbits) - 1;
            int cval = 0;
            if (cbits) cval = symbol(br, st.books[g.class_master[pc]]);
            for (int k = 0; k < cdim; k++) {
                const int book = g.sub_books[pc][cval & csub];
                cval >>= cbits;
                Y[off++] = book >= 0 ? (int16_t)symbol(br, st.books[(size_t)book]) : (int16_t)0;
            }
        }
        if (br.invalid) { zero[i] = true; continue; }
        bool step2[256];
        step2[0] = step2[1] = true;
        for (int j = 2; j < g.values; j++) {
            const int lo = g.lo[j], hi = g.hi[j];
            int pred;
            {   // predict_point, :1446-1455
                const int dy = Y[hi] - Y[lo], adx = g.x[hi] - g.x[lo];
                const int e = std::abs(dy) * (g.x[j] - g.x[lo]);
                const int o = e / adx;
                pred = dy < 0 ? Y[lo] - o : Y[lo] + o;
            }
            const int val = Y[j], highroom = range - pred, lowroom = pred;
            const int room = (highroom < lowroom ? highroom : lowroom) * 2;
            if (val) {
                step2[lo] = step2[hi] = step2[j] = true;
                if (val >= room) Y[j] = (int16_t)(highroom > lowroom ? val - lowroom + pred : pred - val + highroom - 1);
                else Y[j] = (int16_t)((val & 1) ? pred - ((val + 1) >> 1) : pred + (val >> 1));
            } else {
                step2[j] = false;
                Y[j] = (int16_t)pred;
            }
        }
        for (int j = 0; j < g.values; j++)
            if (!step2[j]) Y[j] = -1;
    }
    std::memcpy(really_zero, zero, sizeof(bool) * (size_t)C);
    for (int i = 0; i < map.coupling; i++)
        if (!zero[map.mag[i]] || !zero[map.ang[i]]) zero[map.mag[i]] = zero[map.ang[i]] = false;

    // how far into a channel's spectrum any residue of this packet can write: what AFG_VORBIS_NZ_EIGHTHS declares.  Nothing
    // else puts a nonzero there: inverse coupling of (+0, +0) is (+0, +0) and the floor multiplies
    uint32_t nz_bins = 0;
    // residues (:1586-1713)
    for (int sm = 0; sm < map.submaps; sm++) {
        float *buf[16];
        bool skip[16];
        int ch = 0;
        for (int j = 0; j < C; j++)
            if (map.mux[j] == sm) {
                skip[ch] = zero[j];
                buf[ch] = zero[j] ? nullptr : sc.spec.data() + (size_t)j * (size_t)n;
                ch++;
            }
        const Residue &r = st.residues[map.residue_of[sm]];
        const Book &cb = st.books[(size_t)r.classbook];
        const int classwords = cb.dim;
        const uint32_t actual = r.type == 2 ? (uint32_t)n2 * 2 : (uint32_t)n2;
        const uint32_t rb = std::min(r.begin, actual), re = std::min(r.end, actual);
        const int part_read = (int)((re - rb) / r.part_size);
        if (classwords <= 0) continue;
        if (ch) {
            // type 2 interleaves its channels: position z belongs to bin z / ch
            // (a type-2 vector is not cut at the end of its partition, :1405-1440: up to dim - 1 positions more)
            const uint32_t last = rb + (uint32_t)part_read * r.part_size + (r.type == 2 ? (uint32_t)r.max_dim - 1 : 0u);
            nz_bins = std::max(nz_bins, r.type == 2 ? (last + (uint32_t)ch - 1) / (uint32_t)ch : last);
        }
        const size_t cls_pitch = (size_t)(part_read + classwords + 1);
        std::vector<int> &cls = sc.cls;                          // reused across packets: no allocation on the packet path
        cls.assign((size_t)std::max(ch, 1) * cls_pitch, 0);
        bool done = false;
        if (r.type == 2 && ch != 1) {
            bool any = false;
            for (int j = 0; j < ch; j++) any = any || !skip[j];
            if (!any) continue;
            for (int pass = 0; pass < 8 && !done; pass++) {
                int pcount = 0;
                while (pcount < part_read && !done) {
                    int z = (int)r.begin + pcount * (int)r.part_size;
                    int ci = z % ch, pi = z / ch;
                    if (pass == 0) {
                        int q = symbol(br, cb);
                        if (q < 0) { done = true; break; }
                        for (int k = classwords - 1; k >= 0; k--) {
                            cls[(size_t)(pcount + k)] = q % r.classifications;
                            q /= r.classifications;
                        }
                    }
                    for (int i = 0; i < classwords && pcount < part_read; i++, pcount++) {
                        const int b = r.books[cls[(size_t)pcount]][pass];
                        if (b >= 0) {
                            // codebook_decode_deinterleave_repeat, :1389-1444
                            const Book &bk = st.books[(size_t)b];
                            if (bk.lookup == 0) { done = true; break; }
                            int total = (int)r.part_size, eff = bk.dim;
                            while (total > 0) {
                                float last = 0;
                                const int zz = symbol(br, bk);
                                if (zz < 0) { done = true; break; }
                                if (ci + pi * ch + eff > n2 * ch) eff = n2 * ch - (pi * ch - ci);
                                const float *m = bk.mult.data() + (size_t)bk.vec(zz) * (size_t)bk.dim;
                                for (int e = 0; e < eff; e++) {
                                    const float val = m[e] + last;
                                    if (buf[ci]) buf[ci][pi] += val;
                                    if (++ci == ch) { ci = 0; ++pi; }
                                    if (bk.sequence) last = val;
                                }
                                total -= eff;
                            }
                            if (done) break;
                        } else {
                            z = (int)r.begin + pcount * (int)r.part_size + (int)r.part_size;
                            ci = z % ch;
                            pi = z / ch;
                        }
                    }
                }
            }
            continue;
        }
        for (int pass = 0; pass < 8 && !done; pass++) {
            int pcount = 0;
            while (pcount < part_read && !done) {
                if (pass == 0) {
                    for (int j = 0; j < ch && !done; j++) {
                        if (skip[j]) continue;
                        int q = symbol(br, cb);
                        if (q < 0) { done = true; break; }
                        for (int k = classwords - 1; k >= 0; k--) {
                            cls[(size_t)j * cls_pitch + (size_t)(pcount + k)] = q % r.classifications;
                            q /= r.classifications;
                        }
                    }
                    if (done) break;
                }
                for (int i = 0; i < classwords && pcount < part_read && !done; i++, pcount++) {
                    for (int j = 0; j < ch && !done; j++) {
                        if (skip[j]) continue;
                        const int b = r.books[cls[(size_t)j * cls_pitch + (size_t)pcount]][pass];
                        if (b < 0) continue;
                        const Book &bk = st.books[(size_t)b];
                        float *target = buf[j];
                        int offset = (int)r.begin + pcount * (int)r.part_size;
                        const int np = (int)r.part_size;
                        if (r.type == 0) {
                            const int step = np / bk.dim;
                            for (int k = 0; k < step; k++)
                                if (!add_vector_strided(br, bk, target + offset + k, np - offset - k, step)) { done = true; break; }
                        } else {
                            for (int k = 0; k < np;) {
                                if (!add_vector(br, bk, target + offset, np - k)) { done = true; break; }
                                k += bk.dim;
                                offset += bk.dim;
                            }
                        }
                    }
                }
            }
        }
    }

    if (mode.blockflag) {
        const uint32_t eighth = (uint32_t)n2 / 8;
        flags |= AFG_VORBIS_NZ_EIGHTHS(std::min<uint32_t>(8, (nz_bins + eighth - 1) / eighth));
    }
    if (sc.device_floor) {
        // the tail below as records: the floor points do_floor walks (:2262-2272), in its order, y scaled
        sc.mapping = mode.mapping;
        sc.n_points.assign((size_t)C, 0);
        sc.points.clear();
        for (int i = 0; i < C; i++) {
            if (really_zero[i]) continue;
            const Floor1 &g = st.floors[map.floor_of[map.mux[i]]];
            const int16_t *Y = sc.y.data() + (size_t)i * 256;
            sc.points.push_back(0);
            sc.points.push_back(Y[0] * g.multiplier);
            uint32_t np = 1;
            for (int q = 1; q < g.values; q++) {
                const int j = g.order[q];
                if (Y[j] >= 0) {
                    sc.points.push_back(g.x[j]);
                    sc.points.push_back(Y[j] * g.multiplier);
                    np++;
                }
            }
            sc.n_points[(size_t)i] = np;
        }
        return true;
    }
    // inverse coupling (:2493-2514)
    for (int i = map.coupling - 1; i >= 0; --i) {
        float *m = sc.spec.data() + (size_t)map.mag[i] * (size_t)n, *a = sc.spec.data() + (size_t)map.ang[i] * (size_t)n;
        for (int j = 0; j < n2; j++) {
            float a2, m2;
            if (m[j] > 0) {
                if (a[j] > 0) { m2 = m[j]; a2 = m[j] - a[j]; }
                else { a2 = m[j]; m2 = m[j] + a[j]; }
            } else {
                if (a[j] > 0) { m2 = m[j]; a2 = m[j] + a[j]; }
                else { a2 = m[j]; m2 = m[j] - a[j]; }
            }
            m[j] = m2;
            a[j] = a2;
        }
    }
    // floor curves (:2255-2284)
    for (int i = 0; i < C; i++) {
        float *t = sc.spec.data() + (size_t)i * (size_t)n;
        if (really_zero[i]) { std::fill(t, t + n2, 0.0f); continue; }
        const Floor1 &g = st.floors[map.floor_of[map.mux[i]]];
        const int16_t *Y = sc.y.data() + (size_t)i * 256;
        int lx = 0, ly = Y[0] * g.multiplier;
        for (int q = 1; q < g.values; q++) {
            const int j = g.order[q];
            if (Y[j] >= 0) {
                const int hy = Y[j] * g.multiplier, hx = g.x[j];
                if (lx != hx) render_line(t, lx, ly, hx, hy, n2);
                lx = hx;
                ly = hy;
            }
        }
        if (lx < n2) {
            const float v = bits_f32(k_inverse_db_bits[ly & 255]);
            for (int j = lx; j < n2; j++) t[j] *= v;
        }
    }
    return true;
}

// ---- stream length: the granule position of the last page (:3797-3868) --------------------------------------
uint32_t crc_table[256];
bool crc_ready = false;
void crc_init()
{
    if (crc_ready) return;
    for (uint32_t i = 0; i < 256; i++) {
        uint32_t s = i << 24;
        for (int j = 0; j < 8; ++j) s = (s << 1) ^ (s >= (1U << 31) ? 0x04c11db7u : 0);
        crc_table[i] = s;
    }
    crc_ready = true;
}
// first CRC-valid page at or after `from`: its start, end and last-page flag
bool find_page(const uint8_t *d, size_t n, size_t from, size_t &start, size_t &end, bool &last)
{
    crc_init();
    for (size_t p = from; p + 27 <= n; p++) {
        if (d[p] != 0x4f) continue;
        // `if (retry_loc - 25 > f.stream_len) return 0` (stb_vorbis2.d:3407) with retry_loc = p + 1, unsigned: an 'O' in the
        // first 24 bytes of the data ends the search -- the length scan of a short file whose headers end inside a page
        // (first_audio_page_offset = 0) starts there and so reports no length
        if (p < 24) return false;
        if (std::memcmp(d + p, "OggS", 4) || d[p + 4] != 0) continue;
        const int nseg = d[p + 26];
        if (p + 27 + (size_t)nseg > n) return false;
        size_t len = 0;
        for (int i = 0; i < nseg; i++) len += d[p + 27 + i];
        if (p + 27 + (size_t)nseg + len > n) { if (len) return false; }
        uint32_t crc = 0;
        for (size_t i = 0; i < 27 + (size_t)nseg + len; i++) {
            const uint8_t byte = (i >= 22 && i < 26) ? 0 : d[p + i];
            crc = (crc << 8) ^ crc_table[byte ^ (crc >> 24)];
        }
        if (crc == rd32(d + p + 22)) {
            start = p;
            end = p + 27 + (size_t)nseg + len;
            last = (d[p + 5] & 4) != 0;
            return true;
        }
    }
    return false;
}
uint32_t stream_length(const uint8_t *d, size_t n, size_t first_audio_page)
{
    size_t from = (n >= 65536 && n - 65536 >= first_audio_page) ? n - 65536 : first_audio_page;
    size_t start, end;
    bool last;
    if (!find_page(d, n, from, start, end, last)) return 0;
    size_t last_start = start;
    while (!last) {
        size_t s2, e2;
        if (!find_page(d, n, end, s2, e2, last)) break;
        last_start = s2;
        end = e2;
    }
    const uint32_t lo = rd32(d + last_start + 6), hi = rd32(d + last_start + 10);
    if (lo == 0xffffffffu && hi == 0xffffffffu) return 0;
    return hi ? 0xfffffffeu : lo;
}

}  // namespace

namespace {

// Everything up to and including the setup header: the part of parse_file that can reject the stream.
bool open_stream(const uint8_t *data, size_t size, File &f, Demux &dm, Setup &st)
{
    if (!data || size < 58 || std::memcmp(data, "OggS", 4)) return false;
    demux(data, size, dm);
    if (dm.packets.size() < 3) return false;
    // identification header: alone on the first page, 30 bytes (:2678-2731)
    {
        if (!(data[5] & 2) || (data[5] & 4) || (data[5] & 1) || data[26] != 1 || data[27] != 30) return false;
        const Packet &p = dm.packets[0];
        if (!p.complete || p.n_pieces != 1 || dm.pieces[p.first_piece].len != 30) return false;
        const uint8_t *h = data + dm.pieces[p.first_piece].off;
        if (h[0] != 1 || std::memcmp(h + 1, "vorbis", 6) || rd32(h + 7) != 0) return false;
        f.channels = h[11];
        f.sample_rate = rd32(h + 12);
        if (!f.channels || f.channels > 16 || !f.sample_rate) return false;
        const int log0 = h[28] & 15, log1 = h[28] >> 4;
        if (log0 < 6 || log0 > 13 || log1 < 6 || log1 > 13 || log0 > log1) return false;
        f.blocksize0 = 1 << log0;
        f.blocksize1 = 1 << log1;
        if (!(h[29] & 1)) return false;
        // What the transform stage cannot take is refused here, per file (a plan that fails would fail its whole chunk of
        // the batch): block sizes 64 / 128 (the reference's inverse_mdct is wrong for them, stb_vorbis2.d:2053-2090).
        // (Until round 5 also streams of 6+ channels at block size 8192: the general path kept all channels of a segment
        // in one workgroup's LDS; it is one wavefront per channel now.)
        if (f.blocksize0 < 256) return false;
    }
    st.channels = f.channels;
    st.rate = f.sample_rate;
    st.bs[0] = f.blocksize0;
    st.bs[1] = f.blocksize1;
    {   // comment header (:2736-2786): type 3, "vorbis", vendor, comments, framing bit.  The reference reads it with
        // get8_packet, which answers EOP (-1) past the end of the packet, and get32_packet, which adds four such answers up
        // (:1199-1207): lengths that a damaged byte has made longer than the packet run into the end, the framing "byte" read
        // there is 0xff, and the header passes.  What fails is an allocation: setup_malloc of a size that is not positive
        // (a length of -1 or less -- which is also what four EOPs add up to).  Sizes up to 2 GB are taken to succeed.
        const Packet &p = dm.packets[1];
        if (!p.complete) return false;
        Bits br(data, dm.pieces.data() + p.first_piece, p.n_pieces);
        auto g8 = [&]() -> int { return br.exhausted() ? -1 : (int)br.get(8); };
        auto g32 = [&]() -> int32_t {
            uint32_t x = (uint32_t)g8();
            x += (uint32_t)g8() << 8;
            x += (uint32_t)g8() << 16;
            x += (uint32_t)g8() << 24;
            return (int32_t)x;
        };
        auto allocates = [](int32_t sz) { return (int32_t)(((uint32_t)sz + 3u) & ~3u) > 0; };      // setup_malloc (:556-567)
        auto skip = [&](int32_t len) { for (int32_t i = 0; i < len && !br.exhausted(); i++) br.get(8); };
        if (g8() != 3) return false;
        for (const char *c = "vorbis"; *c; c++)
            if ((uint8_t)g8() != (uint8_t)*c) return false;
        int32_t len = g32();
        if (!allocates((int32_t)((uint32_t)len + 1u))) return false;
        skip(len);
        const int32_t ncomm = g32();
        if (ncomm > 0 && !allocates((int32_t)(8u * (uint32_t)ncomm))) return false;
        for (int32_t k = 0; k < ncomm; k++) {
            len = g32();
            if (!allocates((int32_t)((uint32_t)len + 1u))) return false;
            skip(len);
        }
        if (!((uint8_t)g8() & 1)) return false;
    }
    {   // setup header
        const Packet &p = dm.packets[2];
        Bits br(data, dm.pieces.data() + p.first_piece, p.n_pieces);
        if (br.get(8) != 5) return false;
        for (const char *c = "vorbis"; *c; c++)
            if (br.get(8) != (uint8_t)*c) return false;
        if (!read_setup(br, st)) return false;
        if (!p.complete) return false;
    }
    f.fl_steps.clear();
    for (Mapping &m : st.mappings) {
        m.step_off = (uint32_t)(f.fl_steps.size() / 2);
        for (int i = m.coupling - 1; i >= 0; --i) {
            f.fl_steps.push_back(m.mag[i]);
            f.fl_steps.push_back(m.ang[i]);
        }
    }
    return true;
}

}  // namespace

// Upper bound of the spectrum floats parse_file_into() records, or 0 when parse_file() would reject the stream:
// every audio packet a long block on every channel.
size_t max_spec_floats(const uint8_t *data, size_t size)
{
    File f;
    Demux dm;
    Setup st;
    if (!open_stream(data, size, f, dm, st)) return 0;
    return (dm.packets.size() - 3 + 1) * (size_t)f.channels * (size_t)(f.blocksize1 / 2);
}

bool parse_file(const uint8_t *data, size_t size, File &f, bool device_floor) { return parse_file_into(data, size, f, nullptr, 0, device_floor); }

namespace {

// The packet loop of the pull API with the reference's position bookkeeping (:2531-2596), one packet per call, so that
// a whole file (parse_file_into) and a chunked stream (Reader) run the same code.
struct Walk {
    const uint8_t *data = nullptr;
    size_t size = 0;
    Demux dm;
    Setup st;
    Scratch sc;
    bool first = true, loc_valid = false;
    uint32_t cur_loc = 0;
    int deferred = 0, prev_overlap = 0;
    size_t k = 3;                                          // next packet (0..2 are the headers)
    // the last packet recorded: a chunked stream repeats it at the head of the next chunk (the transform overlaps a
    // block with its predecessor, which the device re-derives from the predecessor's spectrum)
    std::vector<float> last_spec;
    unsigned last_flags = 0;
    bool have_last = false;
    int last_mapping = 0;
    std::vector<uint32_t> last_n_points;
    std::vector<int32_t> last_points;

    // one packet's floor records appended to `f` (its spectrum starts at f.n_spec)
    void add_floor_records(File &f, int n2, int mapping, const std::vector<uint32_t> &n_points, const std::vector<int32_t> &points) const
    {
        afg_vorbis_floor_packet r;
        std::memset(&r, 0, sizeof(r));
        r.spec_off = f.n_spec;
        r.n2 = (uint32_t)n2;
        r.channels = (uint32_t)f.channels;
        r.curve_index = (uint32_t)f.fl_curves.size();
        r.step_off = st.mappings[(size_t)mapping].step_off;
        r.n_steps = (uint32_t)st.mappings[(size_t)mapping].coupling;
        f.fl_packets.push_back(r);
        uint32_t at = (uint32_t)(f.fl_points.size() / 2);
        for (uint32_t np : n_points) {
            f.fl_curves.push_back(afg_vorbis_floor_curve{ at, np });
            at += np;
        }
        f.fl_points.insert(f.fl_points.end(), points.begin(), points.end());
    }

    // Decodes packet k into `f`; false: the stream ends here (nothing recorded).  `recorded` tells whether a record was
    // appended (non-audio packets are skipped).
    bool step(File &f, bool &recorded, bool keep_last)
    {
        recorded = false;
        if (k >= dm.packets.size()) return false;
        const Packet &pk = dm.packets[k];
        unsigned flags = 0;
        int n = 0;
        bool audio = true;
        if (!decode_packet(data, dm, pk, st, sc, flags, n, audio)) return false;
        if (!audio) {
            if (!pk.complete) return false;
            k++;
            return true;
        }
        const int n2 = n >> 1;
        int left_start, right_start, right_end;
        if ((flags & AFG_VORBIS_LONG) && !(flags & AFG_VORBIS_PREV)) left_start = (n - f.blocksize0) >> 2;
        else left_start = 0;
        if ((flags & AFG_VORBIS_LONG) && !(flags & AFG_VORBIS_NEXT)) {
            right_start = (n * 3 - f.blocksize0) >> 2;
            right_end = (n * 3 + f.blocksize0) >> 2;
        } else {
            right_start = n2;
            right_end = n;
        }
        {   // The transform stage overlaps a block's left window with the previous block's right window: they must have
            // the same length, which is what consistent prev/next window flags guarantee.  A stream that breaks this
            // (the reference mixes windows of different lengths then, :2618-2627) ends at that packet here.
            const int left_end = ((flags & AFG_VORBIS_LONG) && !(flags & AFG_VORBIS_PREV)) ? (n + f.blocksize0) >> 2 : n2;
            if (prev_overlap && prev_overlap != left_end - left_start) return false;
            prev_overlap = right_end - right_start;
        }
        int left = left_start, len = right_end;
        bool len_set = false;
        if (first) {
            cur_loc = 0u - (uint32_t)n2;
            deferred = n - right_end;
            loc_valid = true;
        } else if (deferred) {
            if (deferred >= right_start - left_start) {
                deferred -= right_start - left_start;
                left = right_start;
            } else {
                left += deferred;
                deferred = 0;
            }
        }
        if (pk.complete && pk.ends_page_run) {
            if (loc_valid && pk.on_last_page) {
                const uint32_t cur_end = pk.granule_lo;
                if (cur_end < cur_loc + (uint32_t)(right_end - left)) {
                    len = cur_end < cur_loc ? 0 : (int)(cur_end - cur_loc);
                    len += left;
                    if (len > right_end) len = right_end;
                    cur_loc += (uint32_t)len;
                    len_set = true;
                }
            }
            if (!len_set) {
                cur_loc = pk.granule_lo - (uint32_t)(n2 - left);
                loc_valid = true;
            }
        }
        if (!len_set && loc_valid) cur_loc += (uint32_t)(right_start - left);
        // record
        if (f.ext_spec && f.n_spec + (size_t)f.channels * (size_t)n2 > f.ext_cap) { f.overflow = true; return false; }
        f.pflags.push_back((uint8_t)flags);
        if (sc.device_floor) add_floor_records(f, n2, sc.mapping, sc.n_points, sc.points);
        if (f.ext_spec) {
            for (int c = 0; c < f.channels; c++)
                std::memcpy(f.ext_spec + f.n_spec + (size_t)c * (size_t)n2, sc.spec.data() + (size_t)c * (size_t)n, (size_t)n2 * sizeof(float));
        } else {
            for (int c = 0; c < f.channels; c++)
                f.spec.insert(f.spec.end(), sc.spec.begin() + (size_t)c * (size_t)n, sc.spec.begin() + (size_t)c * (size_t)n + (size_t)n2);
        }
        if (keep_last) {
            last_spec.resize((size_t)f.channels * (size_t)n2);
            for (int c = 0; c < f.channels; c++)
                std::memcpy(last_spec.data() + (size_t)c * (size_t)n2, sc.spec.data() + (size_t)c * (size_t)n, (size_t)n2 * sizeof(float));
            last_flags = flags;
            have_last = true;
            if (sc.device_floor) {
                last_mapping = sc.mapping;
                last_n_points = sc.n_points;
                last_points = sc.points;
            }
        }
        f.n_spec += (size_t)f.channels * (size_t)n2;
        int r = std::min(right_start, len);
        int count = first ? 0 : std::max(0, r - left);
        f.take_from.push_back(first ? 0 : left - left_start);
        f.take_count.push_back(count);
        f.pcm_frames += (uint64_t)count;
        first = false;
        recorded = true;
        k++;
        if (!pk.complete) k = dm.packets.size();            // the data ended inside this packet: nothing follows
        return true;
    }
};

}  // namespace

bool parse_file_into(const uint8_t *data, size_t size, File &f, float *spec_dst, size_t cap, bool device_floor)
{
    f = File();
    Walk w;
    if (!open_stream(data, size, f, w.dm, w.st)) { f = File(); return false; }
    w.data = data;
    w.size = size;
    f.device_floor = w.sc.device_floor = device_floor;
    f.ext_spec = spec_dst;
    f.ext_cap = cap;
    bool recorded = false;
    while (w.step(f, recorded, false)) {}
    f.total_samples = stream_length(data, size, w.dm.first_audio_page);
    return true;
}

// ---- chunked reading (the AudioStream surface decodes as the caller pulls, stream.d:429-637) ----
struct Reader::Impl {
    Walk w;
    File meta;
};

Reader::Reader() : p(new Impl) {}
Reader::~Reader() { delete p; }

bool Reader::open(const uint8_t *data, size_t size, File &meta, bool device_floor)
{
    *p = Impl();
    p->meta = File();
    if (!open_stream(data, size, p->meta, p->w.dm, p->w.st)) return false;
    p->w.data = data;
    p->w.size = size;
    p->meta.device_floor = p->w.sc.device_floor = device_floor;
    p->meta.total_samples = stream_length(data, size, p->w.dm.first_audio_page);
    meta = p->meta;
    return true;
}

bool Reader::more(File &out, int max_packets)
{
    out = File();
    out.channels = p->meta.channels;
    out.blocksize0 = p->meta.blocksize0;
    out.blocksize1 = p->meta.blocksize1;
    out.sample_rate = p->meta.sample_rate;
    out.total_samples = p->meta.total_samples;
    out.device_floor = p->meta.device_floor;
    out.fl_steps = p->meta.fl_steps;
    Walk &w = p->w;
    if (w.have_last) {                                     // the predecessor of this chunk's first packet: primes the overlap only
        out.pflags.push_back((uint8_t)w.last_flags);
        if (out.device_floor) {
            const int n2 = ((w.last_flags & AFG_VORBIS_LONG) ? out.blocksize1 : out.blocksize0) >> 1;
            w.add_floor_records(out, n2, w.last_mapping, w.last_n_points, w.last_points);
        }
        out.spec = w.last_spec;
        out.n_spec = out.spec.size();
        out.take_from.push_back(0);
        out.take_count.push_back(0);
    }
    int got = 0;
    bool recorded = false;
    while (got < max_packets && w.step(out, recorded, true))
        if (recorded) got++;
    return got > 0;
}

}  // namespace afg_vorbis
