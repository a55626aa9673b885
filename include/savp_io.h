/* savp_io.h -- C ABI of the input pipeline (libsavp_io.so, host C++).
 *
 * Replaces, for the BAIR / softmotion record layout (one feature per frame) and the KTH layout (one bytes_list per sequence), what the reference builds from TensorFlow's C++ runtime:
 *   tf.data.TFRecordDataset(filenames, buffer_size=8 MiB)            video_prediction/datasets/base_dataset.py:135
 *   shuffle_and_repeat(buffer_size=1024, count=num_epochs) / repeat  base_dataset.py:137-140
 *   tf.parse_single_example(FixedLenFeature([1], tf.string) per frame, FixedLenFeature(shape, tf.float32) for states /
 *   actions)                                                         base_dataset.py:314-345
 *   tf.decode_raw(uint8) + reshape                                   base_dataset.py:158-166 (jpeg_encoding False: softmotion_dataset.py:56-58)
 *   slice_sequences: time_shift / frame_skip sub-sequence sampling   base_dataset.py:189-229
 *   map_and_batch(drop_remainder=True) + prefetch                    base_dataset.py:148-150
 * The uint8 -> float32 [0,1] conversion (tf.image.convert_image_dtype, base_dataset.py:187) and the transpose to time-major
 * run on the GPU (savp_u8_frames_to_f32 in savp_hip.h) so that only 1 byte per value crosses PCIe.
 *
 * TFRecord framing and tf.train.Example are the published formats of the un-vendored dependency tensorflow-gpu>=1.9.0
 * (tensorflow/core/lib/io/record_writer.cc, tensorflow/core/example/{example,feature}.proto); restated, not linked.
 * All functions return 0 on success or a negative SAVP_IO_* code; no exceptions cross the boundary.
 */
#ifndef SAVP_IO_H
#define SAVP_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { SAVP_IO_OK = 0, SAVP_IO_EINVAL = -1, SAVP_IO_EIO = -2, SAVP_IO_ECORRUPT = -3, SAVP_IO_EOF = -4, SAVP_IO_EMISSING = -5 };

/* CRC-32C (Castagnoli) and TFRecord's masked form ((crc >> 15 | crc << 17) + 0xa282ead8). */
uint32_t savp_io_crc32c(const void* data, uint64_t n);
uint32_t savp_io_masked_crc32c(const void* data, uint64_t n);

/* ---- record level ------------------------------------------------------------------------------------------------------- */
typedef struct SavpTfrFile SavpTfrFile;
int savp_tfr_open(const char* path, int64_t buffer_bytes, SavpTfrFile** out);
/* next record: *data points into an internal buffer valid until the next call; SAVP_IO_EOF at the end; CRCs are verified */
int savp_tfr_next(SavpTfrFile* f, const uint8_t** data, uint64_t* len);
void savp_tfr_close(SavpTfrFile* f);

/* ---- tf.train.Example level ---------------------------------------------------------------------------------------------- */
/* Find feature `name` in a serialized Example.  kind: 1 bytes_list, 2 float_list, 3 int64_list.
 * bytes_list: *ptr / *len = the index-th value.  float_list / int64_list: *ptr = packed payload, *len = element count. */
int savp_example_feature(const uint8_t* ex, uint64_t ex_len, const char* name, int32_t index, int32_t* kind,
                         const uint8_t** ptr, uint64_t* len);
/* The index-th value of an int64_list feature (e.g. "sequence_length", "height" of the KTH records, kth_dataset.py:20-24). */
int savp_example_int64(const uint8_t* ex, uint64_t ex_len, const char* name, int32_t index, int64_t* out);
/* Copy a float_list feature (packed or not) into out[0..n); returns SAVP_IO_EINVAL if the element count differs. */
int savp_example_floats(const uint8_t* ex, uint64_t ex_len, const char* name, float* out, int64_t n);

/* ---- batched video pipeline ------------------------------------------------------------------------------------------------ */
typedef struct SavpVideoPipelineArgs {
    const char* const* filenames; int32_t num_files;
    const char* image_key_fmt;      /* e.g. "%d/image_aux1/encoded" (softmotion_dataset.py:36) */
    int32_t example_frames;         /* frames stored per example (30 for BAIR) */
    int32_t height, width, channels;
    int32_t sequence_length;        /* frames per returned sequence */
    int32_t frame_skip, time_shift; /* base_dataset.py:198-214; time_shift 0 = always start at frame 0 */
    int32_t batch_size;
    int32_t shuffle;                /* 1: shuffle file order + 1024-example shuffle buffer (train mode), 0: file order */
    int32_t shuffle_buffer;         /* examples; 0 = 1024 (base_dataset.py:138) */
    int32_t num_epochs;             /* <= 0: repeat forever */
    uint64_t seed;
    int32_t prefetch_batches;       /* depth of the ready queue filled by the reader thread (>= 1) */
    const char* const* float_keys_fmt; const int32_t* float_dims; const int32_t* float_per_frame_minus; int32_t num_float_keys;
                                    /* optional state-like (minus 0) / action-like (minus 1) float features, e.g.
                                       "%d/endeffector_pos" dim 3, "%d/action" dim 4 minus 1 (softmotion_dataset.py:38-40) */
    int32_t var_len;                /* 1: VarLenFeatureVideoDataset layout (base_dataset.py:394-453, KTH): image_key_fmt names ONE
                                       bytes_list feature holding every frame of the sequence, int64 feature "sequence_length" gives its
                                       length; examples shorter than sequence_length are dropped (filter, :401-407); example_frames unused */
} SavpVideoPipelineArgs;
typedef struct SavpVideoPipeline SavpVideoPipeline;
int savp_pipeline_create(const SavpVideoPipelineArgs* a, SavpVideoPipeline** out);
/* Blocks until a batch is ready.  images: uint8 [batch, sequence_length, H, W, C] (caller-owned, e.g. pinned host memory);
 * floats[k]: float32 [batch, sequence_length - minus_k, dim_k * (frame_skip + 1 if action-like else 1)] or NULL.
 * Returns SAVP_IO_EOF when num_epochs are exhausted (the incomplete last batch is dropped: drop_remainder=True). */
int savp_pipeline_next(SavpVideoPipeline* p, uint8_t* images, float* const* floats);
const char* savp_pipeline_error(SavpVideoPipeline* p);
void savp_pipeline_destroy(SavpVideoPipeline* p);

#ifdef __cplusplus
}
#endif
#endif
