// Training-data input path (include/faststyle_io.h): TFRecord framing + tf.train.Example lookup on the
// host (plain C++, no HIP calls) and the TF1 bicubic resize as a device kernel.
//
// Reference: datapipe.py:14-49 (TFRecordReader, parse_single_example, resize_images(method=2)),
// tfrecords_writer.py:217-239 (TFRecordWriter).  The record framing and the Example wire format are
// TensorFlow's published formats (tensorflow/core/lib/io/record_writer.cc, core/example/*.proto); the
// resize restates tensorflow/core/kernels/resize_bicubic_op.cc @ r1.0.
#include "../../include/faststyle_io.h"

#include <cstring>

#include "fs_kernels.h"

namespace fs {

// ---------------------------------------------------------------- CRC-32C
static uint32_t g_crc_tab[8][256];
static bool g_crc_init = false;
static void crc_init() {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        g_crc_tab[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int t = 1; t < 8; ++t) g_crc_tab[t][i] = (g_crc_tab[t - 1][i] >> 8) ^ g_crc_tab[0][g_crc_tab[t - 1][i] & 0xFF];
    g_crc_init = true;
}

// slice-by-8 table walk (portable path)
static uint32_t crc32c_sw(uint32_t c, const unsigned char* p, size_t n) {
    if (!g_crc_init) crc_init();
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = g_crc_tab[7][v & 0xFF] ^ g_crc_tab[6][(v >> 8) & 0xFF] ^ g_crc_tab[5][(v >> 16) & 0xFF] ^
            g_crc_tab[4][(v >> 24) & 0xFF] ^ g_crc_tab[3][(v >> 32) & 0xFF] ^ g_crc_tab[2][(v >> 40) & 0xFF] ^
            g_crc_tab[1][(v >> 48) & 0xFF] ^ g_crc_tab[0][(v >> 56) & 0xFF];
        p += 8;
        n -= 8;
    }
    while (n--) c = g_crc_tab[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c;
}

#if defined(__x86_64__)
// the host CPUs of an MI355X node have the SSE4.2 crc32 instruction (8 bytes / cycle-ish)
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(uint32_t c, const unsigned char* p, size_t n) {
    uint64_t c64 = c;
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        c64 = __builtin_ia32_crc32di(c64, v);
        p += 8;
        n -= 8;
    }
    c = (uint32_t)c64;
    while (n--) c = __builtin_ia32_crc32qi(c, *p++);
    return c;
}
#endif

static uint32_t crc32c(const void* data, size_t n) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
#if defined(__x86_64__)
    static const bool hw = __builtin_cpu_supports("sse4.2");
    if (hw) return crc32c_hw(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
#endif
    return crc32c_sw(0xFFFFFFFFu, p, n) ^ 0xFFFFFFFFu;
}
static uint32_t mask_crc(uint32_t c) { return ((c >> 15) | (c << 17)) + 0xA282EAD8u; }

// ---------------------------------------------------------------- protobuf wire helpers
struct Cursor {
    const unsigned char* p;
    const unsigned char* end;
    bool ok;
};
static uint64_t varint(Cursor& c) {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
        if (c.p >= c.end) {
            c.ok = false;
            return 0;
        }
        const unsigned char b = *c.p++;
        v |= (uint64_t)(b & 0x7F) << shift;
        if (!(b & 0x80)) return v;
    }
    c.ok = false;
    return 0;
}
// next field: number, wire type; for length-delimited fields sub = the payload; for varint val = value
static bool next_field(Cursor& c, int* field, int* wt, Cursor* sub, uint64_t* val) {
    if (c.p >= c.end) return false;
    const uint64_t tag = varint(c);
    if (!c.ok) return false;
    *field = (int)(tag >> 3);
    *wt = (int)(tag & 7);
    switch (*wt) {
        case 0:
            *val = varint(c);
            return c.ok;
        case 1:
            if (c.end - c.p < 8) return c.ok = false;
            c.p += 8;
            return true;
        case 2: {
            const uint64_t len = varint(c);
            if (!c.ok || (uint64_t)(c.end - c.p) < len) return c.ok = false;
            sub->p = c.p;
            sub->end = c.p + len;
            sub->ok = true;
            c.p += len;
            return true;
        }
        case 5:
            if (c.end - c.p < 4) return c.ok = false;
            c.p += 4;
            return true;
        default:
            return c.ok = false;
    }
}

// Example{1: Features{1: map entry{1: key, 2: Feature{1: BytesList | 2: FloatList | 3: Int64List}}}}
// Returns 0 and the Feature payload of `key`, -1 malformed, -2 absent.
static int find_feature(const void* ex, size_t n, const char* key, Cursor* feature) {
    const size_t klen = strlen(key);
    Cursor top{static_cast<const unsigned char*>(ex), static_cast<const unsigned char*>(ex) + n, true};
    int f, wt;
    Cursor feats{}, entry{}, sub{};
    uint64_t val;
    while (next_field(top, &f, &wt, &feats, &val)) {
        if (f != 1 || wt != 2) continue;
        while (next_field(feats, &f, &wt, &entry, &val)) {
            if (f != 1 || wt != 2) continue;
            bool match = false;
            Cursor value{};
            bool have_value = false;
            while (next_field(entry, &f, &wt, &sub, &val)) {
                if (f == 1 && wt == 2) match = (size_t)(sub.end - sub.p) == klen && memcmp(sub.p, key, klen) == 0;
                if (f == 2 && wt == 2) {
                    value = sub;
                    have_value = true;
                }
            }
            if (!entry.ok) return -1;
            if (match && have_value) {
                *feature = value;
                return 0;
            }
        }
        if (!feats.ok) return -1;
    }
    return top.ok ? -2 : -1;
}

// ---------------------------------------------------------------- TF1 bicubic resize
// One thread per output pixel (3 channels).  Every product and sum is rounded separately, in TF's order
// (`#pragma clang fp contract(off)`: hipcc's default would fuse a*b+c into FMAs), so the result is bit-identical
// to the float32 restatement in oracle/datapipe.py.
__device__ __forceinline__ float bicubic_near(float x) {
#pragma clang fp contract(off)
    // TF's coefficient table entry 2i, x = i/1024:  ((a+2)x - (a+3)) x x + 1, a = -0.75
    float t = 1.25f * x;
    t = t - 2.25f;
    t = t * x;
    t = t * x;
    return t + 1.0f;
}
__device__ __forceinline__ float bicubic_far(float x) {
#pragma clang fp contract(off)
    // entry 2i+1 (x += 1):  ((a x - 5a) x + 8a) x - 4a
    x = x + 1.0f;
    float t = -0.75f * x;
    t = t - (-3.75f);
    t = t * x;
    t = t + (-6.0f);
    t = t * x;
    return t - (-3.0f);
}
__device__ __forceinline__ void bicubic_weights(float scale, int out_loc, int limit, float w[4], int idx[4]) {
#pragma clang fp contract(off)
    const float in_f = scale * (float)out_loc;
    const int in_loc = (int)in_f;  // in_loc >= 0: truncation == floor
    const float delta = in_f - (float)in_loc;
    const int offset = (int)lrintf(delta * 1024.0f);
    const float x0 = (float)offset * (1.0f / 1024.0f), x1 = (float)(1024 - offset) * (1.0f / 1024.0f);  // exact
    w[0] = bicubic_far(x0);
    w[1] = bicubic_near(x0);
    w[2] = bicubic_near(x1);
    w[3] = bicubic_far(x1);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int i = in_loc - 1 + k;
        idx[k] = i < 0 ? 0 : (i > limit - 1 ? limit - 1 : i);
    }
}
__device__ __forceinline__ float interp1d(const float w[4], float v0, float v1, float v2, float v3) {
#pragma clang fp contract(off)
    const float p0 = v0 * w[0], p1 = v1 * w[1], p2 = v2 * w[2], p3 = v3 * w[3];
    float acc = p0 + p1;
    acc = acc + p2;
    return acc + p3;
}

// PB = bytes per source pixel: 3 (packed RGB) or 4 (RGBX -- PIL's own storage of an RGB image, which the host hands over as it is: no repack
// under the interpreter lock; the fourth byte is never read)
template <int PB>
__global__ __launch_bounds__(256) void resize_bicubic_u8_kernel(const unsigned char* __restrict__ src, int H, int W,
                                                                float* __restrict__ dst, int Ho, int Wo, float hs, float ws) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= Ho * Wo) return;
    const int oy = i / Wo, ox = i - oy * Wo;
    float wy[4], wx[4];
    int iy[4], ix[4];
    bicubic_weights(hs, oy, H, wy, iy);
    bicubic_weights(ws, ox, W, wx, ix);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float col[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned char* row = src + ((size_t)iy[r] * W) * PB + c;
            col[r] = interp1d(wx, (float)row[ix[0] * PB], (float)row[ix[1] * PB], (float)row[ix[2] * PB], (float)row[ix[3] * PB]);
        }
        dst[(size_t)i * 3 + c] = interp1d(wy, col[0], col[1], col[2], col[3]);
    }
}

// ---------------------------------------------------------------- frame streaming: u8 <-> f32
// stylize_webcam.py:88-95: the captured u8 frame is fed to the net as float (channel order untouched), the
// output goes through numpy .astype(np.uint8) -- truncation toward zero -- and a B<->R swap (cv2.COLOR_BGR2RGB).
__global__ __launch_bounds__(256) void u8_to_f32_kernel(const unsigned char* __restrict__ src, float* __restrict__ dst, size_t n) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i + 3 < n) {
        const uchar4 v = *reinterpret_cast<const uchar4*>(src + i);
        *reinterpret_cast<float4*>(dst + i) = make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w);
    } else {
        for (size_t k = i; k < n; ++k) dst[k] = (float)src[k];
    }
}
__global__ __launch_bounds__(256) void f32_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, size_t npix,
                                                        int swap_rb) {
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const float r = src[p * 3], g = src[p * 3 + 1], b = src[p * 3 + 2];
    // values are in [0,255] (127.5*tanh+127.5); the clamp only guards the u8 conversion against NaN/garbage
    const unsigned char ur = (unsigned char)fminf(fmaxf(r, 0.f), 255.f), ug = (unsigned char)fminf(fmaxf(g, 0.f), 255.f),
                        ub = (unsigned char)fminf(fmaxf(b, 0.f), 255.f);
    dst[p * 3] = swap_rb ? ub : ur;
    dst[p * 3 + 1] = ug;
    dst[p * 3 + 2] = swap_rb ? ur : ub;
}
int u8_to_f32(const unsigned char* src, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(u8_to_f32_kernel, dim3((unsigned)((n / 4 + 256) / 256)), dim3(256), 0, s, src, dst, n);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}
int f32_to_u8(const float* src, unsigned char* dst, size_t npix, int swap_rb, hipStream_t s) {
    hipLaunchKernelGGL(f32_to_u8_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, src, dst, npix, swap_rb);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

int resize_bicubic_u8(const unsigned char* src, int H, int W, float* dst, int Ho, int Wo, hipStream_t s, int pixel_bytes) {
    // CalculateResizeScale(in, out, align_corners=false) = in / static_cast<float>(out)
    const float hs = (float)H / (float)Ho, ws = (float)W / (float)Wo;
    if (pixel_bytes == 4) hipLaunchKernelGGL(resize_bicubic_u8_kernel<4>, dim3(cdiv(Ho * Wo, 256)), dim3(256), 0, s, src, H, W, dst, Ho, Wo, hs, ws);
    else if (pixel_bytes == 3) hipLaunchKernelGGL(resize_bicubic_u8_kernel<3>, dim3(cdiv(Ho * Wo, 256)), dim3(256), 0, s, src, H, W, dst, Ho, Wo, hs, ws);
    else return -1;
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace fs

extern "C" {

uint32_t fs_crc32c(const void* data, size_t n) { return fs::crc32c(data, n); }
uint32_t fs_crc32c_masked(const void* data, size_t n) { return fs::mask_crc(fs::crc32c(data, n)); }

long long fs_tfrecord_scan(const void* buf, size_t n, int verify_crc, uint64_t* payload_off, uint64_t* payload_len,
                           size_t cap) {
    const unsigned char* p = static_cast<const unsigned char*>(buf);
    size_t pos = 0;
    long long count = 0;
    while (pos < n) {
        if (n - pos < 12) return fs::set_error(-1, "fs_tfrecord_scan: truncated record header at byte %zu", pos);
        uint64_t len;
        uint32_t crc;
        memcpy(&len, p + pos, 8);
        memcpy(&crc, p + pos + 8, 4);
        if (verify_crc && fs::mask_crc(fs::crc32c(p + pos, 8)) != crc)
            return fs::set_error(-2, "fs_tfrecord_scan: length checksum mismatch at byte %zu", pos);
        if (len > n - pos - 12 || n - pos - 12 - len < 4)
            return fs::set_error(-1, "fs_tfrecord_scan: truncated record payload at byte %zu", pos);
        const size_t off = pos + 12;
        if (verify_crc) {
            memcpy(&crc, p + off + len, 4);
            if (fs::mask_crc(fs::crc32c(p + off, (size_t)len)) != crc)
                return fs::set_error(-3, "fs_tfrecord_scan: payload checksum mismatch at byte %zu", pos);
        }
        if ((size_t)count < cap) {
            payload_off[count] = off;
            payload_len[count] = len;
        }
        ++count;
        pos = off + (size_t)len + 4;
    }
    return count;
}

size_t fs_tfrecord_frame(const void* payload, size_t n, void* out) {
    unsigned char* o = static_cast<unsigned char*>(out);
    const uint64_t len = n;
    memcpy(o, &len, 8);
    uint32_t crc = fs::mask_crc(fs::crc32c(o, 8));
    memcpy(o + 8, &crc, 4);
    memcpy(o + 12, payload, n);
    crc = fs::mask_crc(fs::crc32c(payload, n));
    memcpy(o + 12 + n, &crc, 4);
    return n + 16;
}

int fs_example_bytes(const void* ex, size_t n, const char* key, uint64_t* off, uint64_t* len) {
    fs::Cursor feat{};
    const int rc = fs::find_feature(ex, n, key, &feat);
    if (rc) return fs::set_error(rc, "fs_example_bytes: key '%s' %s", key, rc == -2 ? "absent" : "in a malformed Example");
    int f, wt;
    fs::Cursor list{}, v{};
    uint64_t val;
    while (fs::next_field(feat, &f, &wt, &list, &val)) {
        if (f != 1 || wt != 2) continue;  // BytesList
        while (fs::next_field(list, &f, &wt, &v, &val))
            if (f == 1 && wt == 2) {
                *off = (uint64_t)(v.p - static_cast<const unsigned char*>(ex));
                *len = (uint64_t)(v.end - v.p);
                return 0;
            }
    }
    return fs::set_error(-3, "fs_example_bytes: feature '%s' holds no bytes value", key);
}

int fs_example_int64(const void* ex, size_t n, const char* key, long long* value) {
    fs::Cursor feat{};
    const int rc = fs::find_feature(ex, n, key, &feat);
    if (rc) return fs::set_error(rc, "fs_example_int64: key '%s' %s", key, rc == -2 ? "absent" : "in a malformed Example");
    int f, wt;
    fs::Cursor list{}, packed{};
    uint64_t val;
    while (fs::next_field(feat, &f, &wt, &list, &val)) {
        if (f != 3 || wt != 2) continue;  // Int64List
        while (fs::next_field(list, &f, &wt, &packed, &val)) {
            if (f != 1) continue;
            if (wt == 0) {  // unpacked repeated int64
                *value = (long long)val;
                return 0;
            }
            if (wt == 2 && packed.p < packed.end) {  // packed
                *value = (long long)fs::varint(packed);
                if (packed.ok) return 0;
            }
        }
    }
    return fs::set_error(-3, "fs_example_int64: feature '%s' holds no int64 value", key);
}

}  // extern "C"
