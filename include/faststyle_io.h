/* faststyle_io.h -- C ABI of the training-data input path.
 *
 * Replaces, for the reference's train.py input pipeline:
 *   datapipe.py:38-49   tf.TFRecordReader().read + tf.parse_single_example   -> fs_tfrecord_scan, fs_example_bytes/_int64
 *   datapipe.py:24      tf.image.resize_images(image, size, method=2)          -> fs_resize_bicubic_u8 (device kernel)
 *   tfrecords_writer.py:217-239  tf.python_io.TFRecordWriter.write             -> fs_tfrecord_frame
 * JPEG entropy decoding stays on the host (libjpeg through PIL, several threads); everything after
 * the decoded u8 pixels runs on the GPU: the image is uploaded as u8 (a quarter of the fp32 bytes over
 * PCIe), resized by the TF1 bicubic kernel straight into the HBM-resident shuffle buffer.
 *
 * The host-side functions are pure C (no HIP calls) and work without a GPU.  All return 0 / a
 * non-negative count on success and a negative code on error (fs_last_error() has the text).
 */
#ifndef FASTSTYLE_IO_H
#define FASTSTYLE_IO_H
#include <stddef.h>
#include <stdint.h>

#include "faststyle_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* CRC-32C (Castagnoli) and TensorFlow's masked form ((crc >> 15 | crc << 17) + 0xa282ead8), the
 * checksum of TFRecord framing (tensorflow/core/lib/io/record_writer.cc) and of bundle checkpoints. */
uint32_t fs_crc32c(const void* data, size_t n);
uint32_t fs_crc32c_masked(const void* data, size_t n);

/* Walks the records of one TFRecord file image held in memory: each record is
 * {uint64 length, uint32 masked_crc(length), payload, uint32 masked_crc(payload)} (little endian).
 * Writes the payload offsets/lengths of the first `cap` records and returns the TOTAL number of
 * records (call with cap = 0 to count).  verify_crc != 0 checks both checksums of every record.
 * Errors: -1 truncated file, -2 length checksum mismatch, -3 payload checksum mismatch. */
long long fs_tfrecord_scan(const void* buf, size_t n, int verify_crc, uint64_t* payload_off, uint64_t* payload_len,
                           size_t cap);

/* Frames one payload for writing: out must hold n + 16 bytes; returns n + 16. */
size_t fs_tfrecord_frame(const void* payload, size_t n, void* out);

/* tf.parse_single_example restricted to what datapipe.py:40-46 asks for: looks `key` up in a
 * serialized tf.train.Example.  _bytes: offset/length (within ex) of bytes_list.value[0];
 * _int64: int64_list.value[0] (packed or unpacked encoding).
 * Errors: -1 malformed proto, -2 key absent, -3 feature has another kind / is empty. */
int fs_example_bytes(const void* ex, size_t n, const char* key, uint64_t* off, uint64_t* len);
int fs_example_int64(const void* ex, size_t n, const char* key, long long* value);

/* tf.image.resize_images(method=2) of TF 1.0 = ResizeBicubic, align_corners=False, legacy (no
 * half-pixel) coordinates, Keys a = -0.75 through TF's 1024-entry coefficient table, borders clamped,
 * result NOT clipped to [0,255] (tensorflow/core/kernels/resize_bicubic_op.cc @ r1.0).
 * src: device u8 [H,W,3]; dst: device f32 [Ho,Wo,3].  Asynchronous on the ctx stream. */
int fs_resize_bicubic_u8(fs_ctx* ctx, const unsigned char* src, int H, int W, float* dst, int Ho, int Wo);
/* The same with pixel_bytes bytes per source pixel: 3 = packed RGB (what fs_resize_bicubic_u8 takes), 4 = RGBX -- the storage a JPEG decoder
 * such as PIL's keeps an RGB image in, handed over without a host-side repack (faststyle_amd/datapipe.py exports it zero-copy through the Arrow C
 * data interface; the repack was the largest interpreter-locked piece of a decode thread's work).  The fourth byte is never read; the result is
 * bit-identical to the packed form's.  Error -2: another pixel size. */
int fs_resize_bicubic_u8x(fs_ctx* ctx, const unsigned char* src, int H, int W, int pixel_bytes, float* dst, int Ho, int Wo);

/* Frame streaming (stylize_webcam.py:88-95): u8 frame -> float net input (channel order untouched), and
 * net output -> u8 by truncation (numpy .astype(np.uint8)) with an optional R<->B swap
 * (cv2.cvtColor(..., COLOR_BGR2RGB)).  src of _u8_to_f32: 4-byte aligned; dst: 16-byte aligned. */
int fs_u8_to_f32(fs_ctx* ctx, const unsigned char* src, size_t n, float* dst);
int fs_f32_to_u8(fs_ctx* ctx, const float* src, size_t npix, int swap_rb, unsigned char* dst);

#ifdef __cplusplus
}
#endif
#endif
