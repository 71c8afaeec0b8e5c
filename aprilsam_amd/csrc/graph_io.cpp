// graph_io.cpp — reader / writer of the reference's `.graph` files.
//
// File format, as written by april_graph_save (aprilsam/april_graph.c:377-396) through the "stype" object stream
// (aprilsam/common/stype.c:75-107, encode_bytes.h:120-250), everything big-endian:
//
//   object  := u64 magic | u32 len(name) | name bytes | u32 len(payload) | payload | u64 magic (same value)
//              a NULL object has name "" and an empty payload
//   file    := object "april_graph_t"
//   graph   := { u8 1 object(node) }* { u8 2 object(factor) }* u8 0 object(attributes or NULL)      (april_graph.c:250-277)
//   "april_graph_node_xyt"      := f64 state[3] | u8 has_init [f64 x3] | u8 has_truth [f64 x3] | object(attributes)   (april_graph_xyt.c:358-383)
//   "april_graph_factor_xyt"    := u32 a | u32 b | f64 z[3] | u8 has_ztruth [f64 x3] | f64 W[9] | object(attributes)  (april_graph_xyt.c:216-240)
//   "april_graph_factor_xytpos" := u32 a |         f64 z[3] | u8 has_ztruth [f64 x3] | f64 W[9] | object(attributes)  (april_graph_xytpos.c:133-156)
//
// The magic of every object comes from ONE process-wide counter that starts at 0x7b287f8a1579a0ed and is bumped by
// every stype_encode_object call, including the calls of the length-measuring passes (an object is encoded twice: once
// into nothing to learn its length, once for real, recursively).  The writer below keeps the same call structure, so a
// file written from a fresh counter carries exactly the magics a fresh reference process would produce; `magic_offset`
// reproduces a process that had already encoded that many objects (data/M3500.graph was saved by the demo after it
// had copied its 5453 factors: offset 8 * 5453).
//
// Attributes ("april_graph_attr_t" := { u8 1 | key | object(value) }* u8 0, april_graph.c:178-212): string values
// ("string" := u32 len | bytes, stype_basic_types.c:65-76) are kept in this library's small attribute store and written
// back in insertion order; values of any other type are stepped over on input (lengths are explicit) with one warning.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/aprilsam_amd.h"

extern "C" {
int aprilsam_amd_attr_put_string(void **attr_slot, const char *key, const char *value);
int aprilsam_amd_attr_count(const void *attr);
int aprilsam_amd_attr_item(const void *attr, int i, const char **key, const char **value);
}

namespace {

constexpr uint64_t MAGIC0 = 0x7b287f8a1579a0edULL;

// ---- output: a sink that either counts or stores, like the reference's data == NULL convention ------------------------
struct Sink {
    std::vector<uint8_t> *buf;    // null: measuring pass
    uint32_t pos = 0;
    void u8(uint8_t v) { if (buf) buf->push_back(v); pos += 1; }
    void u32(uint32_t v) { for (int s = 24; s >= 0; s -= 8) u8((uint8_t)(v >> s)); }
    void u64(uint64_t v) { for (int s = 56; s >= 0; s -= 8) u8((uint8_t)(v >> s)); }
    void f64(double d) { uint64_t v; memcpy(&v, &d, 8); u64(v); }
    void str(const char *s) { uint32_t n = (uint32_t)strlen(s); u32(n); for (uint32_t i = 0; i < n; i++) u8((uint8_t)s[i]); }
};
struct Writer {
    uint64_t next_magic;
    // body: void(Sink&) that may call object() recursively.  Mirrors stype_encode_object's double pass.
    template <class Body> void object(Sink &out, const char *name, Body body) {
        const uint64_t magic = next_magic++;
        out.u64(magic);
        out.str(name);
        Sink measure{ nullptr, 0 };
        body(measure);
        out.u32(measure.pos);
        body(out);
        out.u64(magic);
    }
    void null_object(Sink &out) {
        const uint64_t magic = next_magic++;
        out.u64(magic); out.str(""); out.u32(0); out.u64(magic);
    }
};
static void attr_object(Writer &w, Sink &o, const void *attr) {
    const int n = aprilsam_amd_attr_count(attr);
    if (n == 0) { w.null_object(o); return; }
    w.object(o, "april_graph_attr_t", [&](Sink &b) {
        for (int i = 0; i < n; i++) {
            const char *k, *v;
            aprilsam_amd_attr_item(attr, i, &k, &v);
            b.u8(1); b.str(k);
            w.object(b, "string", [&](Sink &sv) { sv.str(v); });
        }
        b.u8(0);
    });
}
static void opt3(Sink &o, const double *p) { if (p) { o.u8(1); for (int i = 0; i < 3; i++) o.f64(p[i]); } else o.u8(0); }

// ---- input ---------------------------------------------------------------------------------------------------------
struct Reader {
    const uint8_t *d; uint32_t n, pos = 0; bool ok = true;
    bool need(uint32_t k) { if (pos + (uint64_t)k > n) { ok = false; return false; } return true; }
    uint8_t u8() { if (!need(1)) return 0; return d[pos++]; }
    uint32_t u32() { if (!need(4)) return 0; uint32_t v = 0; for (int i = 0; i < 4; i++) v = (v << 8) | d[pos++]; return v; }
    uint64_t u64() { if (!need(8)) return 0; uint64_t v = 0; for (int i = 0; i < 8; i++) v = (v << 8) | d[pos++]; return v; }
    double f64() { uint64_t v = u64(); double x; memcpy(&x, &v, 8); return x; }
    std::string str() { uint32_t k = u32(); if (!need(k)) return ""; std::string s((const char *)d + pos, k); pos += k; return s; }
    bool opt3(double *out) { if (!u8()) return false; for (int i = 0; i < 3; i++) out[i] = f64(); return true; }
    // header of an object: returns false on a framing error; *end = position of the trailing magic
    bool begin(std::string *name, uint32_t *end, uint64_t *magic) {
        *magic = u64(); *name = str(); const uint32_t len = u32();
        if (!ok || !need(len + 8)) { ok = false; return false; }
        *end = pos + len;
        return true;
    }
    bool finish(uint32_t end, uint64_t magic) {       // skips whatever the payload parser left (attributes, unknown types)
        pos = end;
        if (u64() != magic) ok = false;
        return ok;
    }
    void skip_object() { std::string nm; uint32_t end; uint64_t m; if (begin(&nm, &end, &m)) finish(end, m); }
    // object(attributes): string values go to *slot
    void attributes(void **slot, int *foreign) {
        std::string nm; uint32_t end; uint64_t m;
        if (!begin(&nm, &end, &m)) return;
        if (nm == "april_graph_attr_t") {
            while (ok && pos < end && u8() == 1) {
                const std::string key = str();
                std::string vn; uint32_t vend; uint64_t vm;
                if (!begin(&vn, &vend, &vm)) break;
                if (vn == "string") { const std::string v = str(); if (ok) aprilsam_amd_attr_put_string(slot, key.c_str(), v.c_str()); }
                else if (!(*foreign)++) fprintf(stderr, "aprilsam_amd_graph_load: attribute values of type '%s' are not kept\n", vn.c_str());
                if (!finish(vend, vm)) break;
            }
        }
        finish(end, m);
    }
};

}  // namespace

extern "C" {

// april_graph_save (april_graph.c:377-396): 1 on success, 0 on failure — the reference's convention
int aprilsam_amd_graph_save_ex(april_graph_t *g, const char *path, unsigned long long magic_offset) {
    if (!g || !path) return 0;
    april_graph_node_t **ns = (april_graph_node_t **)g->nodes->data;
    april_graph_factor_t **fs = (april_graph_factor_t **)g->factors->data;
    for (int i = 0; i < g->nodes->size; i++)
        if (ns[i]->type != APRIL_GRAPH_NODE_XYT_TYPE) { fprintf(stderr, "aprilsam_amd_graph_save: node %d has type %d, only xyt nodes have a file encoding\n", i, ns[i]->type); return 0; }
    for (int i = 0; i < g->factors->size; i++)
        if (fs[i]->type != APRIL_GRAPH_FACTOR_XYT_TYPE && fs[i]->type != APRIL_GRAPH_FACTOR_XYTPOS_TYPE) {
            fprintf(stderr, "aprilsam_amd_graph_save: factor %d has type %d, only xyt / xytpos factors have a file encoding\n", i, fs[i]->type);
            return 0;
        }
    Writer w{ MAGIC0 + magic_offset };
    auto graph_body = [&](Sink &o) {
        for (int i = 0; i < g->nodes->size; i++) {
            const april_graph_node_t *n = ns[i];
            o.u8(1);
            w.object(o, "april_graph_node_xyt", [&](Sink &b) {
                for (int k = 0; k < 3; k++) b.f64(n->state[k]);
                opt3(b, n->init); opt3(b, n->truth);
                attr_object(w, b, n->attr);
            });
        }
        for (int i = 0; i < g->factors->size; i++) {
            const april_graph_factor_t *f = fs[i];
            o.u8(2);
            const bool binary = f->type == APRIL_GRAPH_FACTOR_XYT_TYPE;
            w.object(o, binary ? "april_graph_factor_xyt" : "april_graph_factor_xytpos", [&](Sink &b) {
                b.u32((uint32_t)f->nodes[0]);
                if (binary) b.u32((uint32_t)f->nodes[1]);
                for (int k = 0; k < 3; k++) b.f64(f->u.common.z[k]);
                opt3(b, f->u.common.ztruth);
                for (int k = 0; k < 9; k++) b.f64(f->u.common.W->data[k]);
                attr_object(w, b, f->attr);
            });
        }
        o.u8(0);
        attr_object(w, o, g->attr);
    };
    // april_graph_save encodes the whole graph twice (length, then data): so does this, for the magics' sake
    Sink measure{ nullptr, 0 };
    w.object(measure, "april_graph_t", graph_body);
    std::vector<uint8_t> buf; buf.reserve(measure.pos);
    Sink out{ &buf, 0 };
    w.object(out, "april_graph_t", graph_body);
    FILE *fp = fopen(path, "wb");
    if (!fp) { fprintf(stderr, "aprilsam_amd_graph_save: cannot open %s\n", path); return 0; }
    const size_t wr = fwrite(buf.data(), 1, buf.size(), fp);
    fclose(fp);
    return wr == buf.size() ? 1 : 0;
}
int aprilsam_amd_graph_save(april_graph_t *g, const char *path) { return aprilsam_amd_graph_save_ex(g, path, 0); }

// april_graph_create_from_file (april_graph.c:398-426): NULL on failure
april_graph_t *aprilsam_amd_graph_load(const char *path) {
    FILE *fp = path ? fopen(path, "rb") : nullptr;
    if (!fp) return nullptr;
    fseek(fp, 0, SEEK_END);
    const long len = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    if (len <= 0 || len > 0x7fffffffL) { fclose(fp); return nullptr; }
    std::vector<uint8_t> data((size_t)len);
    const size_t rd = fread(data.data(), 1, (size_t)len, fp);
    fclose(fp);
    if (rd != (size_t)len) return nullptr;
    Reader r{ data.data(), (uint32_t)len };
    std::string name; uint32_t gend; uint64_t gmagic;
    if (!r.begin(&name, &gend, &gmagic) || name != "april_graph_t") return nullptr;
    april_graph_t *g = april_graph_create();
    int unknown = 0, foreign = 0;
    while (r.ok && r.pos < gend) {
        const uint8_t op = r.u8();
        if (op == 0) break;
        if (op != 1 && op != 2) { r.ok = false; break; }         // the reference asserts here (april_graph.c:307-311)
        uint32_t end; uint64_t magic;
        if (!r.begin(&name, &end, &magic)) break;
        if (op == 1 && name == "april_graph_node_xyt") {
            double st[3], in[3], tr[3];
            for (int k = 0; k < 3; k++) st[k] = r.f64();
            const bool hi = r.opt3(in), ht = r.opt3(tr);
            if (r.ok) {
                april_graph_node_t *nd = april_graph_node_xyt_create(st, hi ? in : nullptr, ht ? tr : nullptr);
                r.attributes(&nd->attr, &foreign);
                aprilsam_amd_graph_add_node(g, nd);
            }
        } else if (op == 2 && (name == "april_graph_factor_xyt" || name == "april_graph_factor_xytpos")) {
            const bool binary = name == "april_graph_factor_xyt";
            const int a = (int)r.u32(), b = binary ? (int)r.u32() : -1;
            double z[3], zt[3];
            for (int k = 0; k < 3; k++) z[k] = r.f64();
            const bool hz = r.opt3(zt);
            struct { unsigned nrows, ncols; double data[9]; } W = { 3, 3, { 0 } };
            for (int k = 0; k < 9; k++) W.data[k] = r.f64();
            if (r.ok) {
                april_graph_factor_t *f = binary ? april_graph_factor_xyt_create(a, b, z, hz ? zt : nullptr, (const matd_t *)&W)
                                                 : april_graph_factor_xytpos_create(a, z, hz ? zt : nullptr, (matd_t *)&W);
                r.attributes(&f->attr, &foreign);
                aprilsam_amd_graph_add_factor(g, f);
            }
        } else if (!unknown++) {
            fprintf(stderr, "aprilsam_amd_graph_load: skipping objects of unknown type '%s'\n", name.c_str());   // stype.c:128-131 prints the same way
        }
        if (!r.finish(end, magic)) break;              // also steps over the object's attributes
    }
    if (r.ok) { r.attributes(&g->attr, &foreign); r.finish(gend, gmagic); }       // graph attributes, trailing magic
    if (!r.ok) { fprintf(stderr, "aprilsam_amd_graph_load: %s is not a well-formed .graph file\n", path); april_graph_destroy(g); return nullptr; }
    return g;
}

// the reference's own names for the two entry points (aprilsam.h; april_graph.c:377-426), and its type registration
// hook, which has nothing to register here
int april_graph_save(april_graph_t *graph, const char *path) { return aprilsam_amd_graph_save_ex(graph, path, 0); }
april_graph_t *april_graph_create_from_file(const char *path) { return aprilsam_amd_graph_load(path); }
void april_graph_stype_init(void) {}

}  // extern "C"
