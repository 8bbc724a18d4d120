/*
 * dietgpu_amd.h -- C ABI of the MI355X-native batched rANS / float codec.
 *
 * This is the drop-in boundary for the dietgpu hot path.  The reference has no
 * C ABI (its public API is C++: dietgpu/ans/GpuANSCodec.h and
 * dietgpu/float/GpuFloatCodec.h, taking `StackDeviceMemory&` and
 * `cudaStream_t`); every entry point below names the reference interface it
 * replaces.  Conventions are the reference's own (GpuANSCodec.h:61-341,
 * GpuFloatCodec.h:103-292):
 *
 *   - `in` / `out` / `inSize` / `outCapacity` are HOST arrays holding DEVICE
 *     pointers / sizes; `*_dev` arguments are device memory.
 *   - sizes are BYTES for the ans_* calls and FLOAT WORDS for the float_* calls.
 *   - `outSize_dev`, `outSuccess_dev`, `histogram_dev` may be NULL.
 *   - everything is enqueued on `stream` (a hipStream_t passed as void*); no
 *     host synchronisation happens unless a checksum has to be verified.
 *   - ANS inputs must be 4-byte aligned (kANSRequiredAlignment); float inputs
 *     float-word aligned; compressed buffers 16-byte aligned.  Nothing more is
 *     needed for speed: uncompressed elements at any such address (elements of
 *     a split tensor, rows of a matrix) take the vector paths of every kernel.
 *   - probBits must be 9, 10 or 11.
 *
 * Temporary memory: the reference threads a `StackDeviceMemory&` through every
 * call (dietgpu/utils/StackDeviceMemory.h:141-157).  Here the caller passes a
 * raw device region (`temp_dev`, `tempBytes`; may be NULL/0).  If it is too
 * small the library falls back to hipMalloc + a stderr warning, exactly like
 * the reference's overflow path (StackDeviceMemory.cpp:119-139).  `tempUsed`
 * (nullable) receives the bytes of temp memory the call needed, which is what
 * `StackDeviceMemory::getMaxMemoryUsage()` reports upstream.  The C++ wrappers
 * in include/dietgpu_amd/ re-create the exact `dietgpu::` signatures on top.
 *
 * Return value: 0 on success, a DGPU_ERR_* code otherwise.  Decode calls
 * additionally report a checksum mismatch as DGPU_ERR_CHECKSUM_MISMATCH with
 * the first failing batch index in `*errBatch`; EVERY failing member (upstream
 * pushes each one into `errorInfo`, GpuANSDecode.cuh:581-590,
 * GpuFloatDecompress.cuh:720-733) is available from
 * dgpu_last_checksum_mismatches() (ANSDecodeStatus / FloatDecompressStatus,
 * GpuANSCodec.h:45-59, GpuFloatCodec.h:84-99).
 */
#ifndef DIETGPU_AMD_H
#define DIETGPU_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGPU_OK 0
#define DGPU_ERR_INVALID_ARGUMENT 1
#define DGPU_ERR_HIP 2
#define DGPU_ERR_CHECKSUM_MISMATCH 3

/* FloatType, dietgpu/float/GpuFloatCodec.h:21-26 */
#define DGPU_FLOAT_UNDEFINED 0u
#define DGPU_FLOAT16 1u
#define DGPU_BFLOAT16 2u
#define DGPU_FLOAT32 3u

/* kANSRequiredAlignment / kANSDefaultProbBits, GpuANSCodec.h:16-20 */
#define DGPU_ANS_REQUIRED_ALIGNMENT 4
#define DGPU_ANS_DEFAULT_PROB_BITS 10

const char* dgpu_version(void);
/* Bumped whenever an entry point of this header is added, removed or changes its meaning.  Code that is built
 * separately against this header (the tensor-op library of this repository, a cgo / JNI binding) compares the value it was compiled with
 * against the library it finds at run time, so that a stale build fails at load instead of inside a call. */
#define DGPU_ABI_VERSION 7u
uint32_t dgpu_abi_version(void);
/* Text of the last error on the calling thread (HIP error string, failed
 * precondition).  The reference aborts through glog CHECK instead. */
const char* dgpu_last_error(void);

/* The batch members whose checksum did not match in the LAST decode call the calling thread made (thread-local,
 * like dgpu_last_error): returns their number and writes up to `cap` of them, ascending batch index, into the
 * arrays that are not NULL.  Replaces the `errorInfo` vector of ANSDecodeStatus / FloatDecompressStatus
 * (GpuANSDecode.cuh:581-590, GpuFloatDecompress.cuh:720-733), which lists every mismatching member. */
uint32_t dgpu_last_checksum_mismatches(int32_t* batchIdx, uint32_t* expected, uint32_t* got, uint32_t cap);

/* ---- size queries -------------------------------------------------------- */
/* getMaxCompressedSize, GpuANSCodec.h:22 / GpuANSEncode.cu:13-25.  Upstream CHECKs the result against INT32_MAX
 * (GpuANSEncode.cu:22), i.e. aborts for inputs of more than 1 717 538 816 bytes; here such a size returns 0 (the
 * C++ mirror aborts like upstream) and the encode entry points reject it with DGPU_ERR_INVALID_ARGUMENT. */
uint32_t dgpu_ans_max_compressed_size(uint32_t uncompressedBytes);
/* getMaxFloatCompressedSize, GpuFloatCodec.h:31 / GpuFloatCompress.cu:23-45 (0 beyond the same guard, which
 * applies to the exponent plane: numFloats bytes) */
uint32_t dgpu_float_max_compressed_size(uint32_t floatType, uint32_t numFloats);
/* Upper bound of temp memory the matching call needs, so a caller can size
 * `temp_dev` once and stay allocation-free (README.md:90-94). */
size_t dgpu_ans_encode_temp_bytes(uint32_t numInBatch, uint32_t maxBytes);
size_t dgpu_ans_decode_temp_bytes(uint32_t numInBatch, uint32_t maxBytes, int probBits);
size_t dgpu_float_compress_temp_bytes(uint32_t floatType, uint32_t numInBatch, uint32_t maxFloats);
size_t dgpu_float_decompress_temp_bytes(uint32_t floatType, uint32_t numInBatch, uint32_t maxFloats, int probBits);

/* ---- rANS encode --------------------------------------------------------- */
/* ansEncodeBatchStride, GpuANSCodec.h:65-98 / GpuANSEncode.cu:27-53 */
int dgpu_ans_encode_batch_stride(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* in_dev, uint32_t inPerBatchSize, uint32_t inPerBatchStride,
    const uint32_t* histogram_dev,
    void* out_dev, uint32_t outPerBatchStride,
    uint32_t* outSize_dev,
    void* stream);

/* ansEncodeBatchPointer, GpuANSCodec.h:100-128 / GpuANSEncode.cu:55-113 */
int dgpu_ans_encode_batch_pointer(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in, const uint32_t* inSize,
    const uint32_t* histogram_dev,
    void* const* out,
    uint32_t* outSize_dev,
    void* stream);

/* ansEncodeBatchSplitSize, GpuANSCodec.h:130-164 / GpuANSEncode.cu:115-179 */
int dgpu_ans_encode_batch_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* in_dev, const uint32_t* inSplitSizes,
    const uint32_t* histogram_dev,
    void* out_dev, uint32_t outStride,
    uint32_t* outSize_dev,
    void* stream);

/* ---- rANS decode --------------------------------------------------------- */
/* What the entry points WITHOUT input sizes may read (the reference's contract, made explicit): up to
 * dgpu_ans_max_compressed_size(outCapacity) / dgpu_float_max_compressed_size(type, outCapacity) bytes from every input
 * pointer -- the size of the buffer an encoder was given for an element of that capacity (GpuANSCodec.h:24-26,
 * GpuFloatCodec.h:54-57).  Inside that extent every format invariant is checked before it is followed; the decoders
 * request the probability table and (float archives, once the float header has been checked) the block descriptor
 * and lane states of the blocks the header implies in the same round trip as the ANS header, i.e. before that header
 * has been validated.  Callers that hold archives in buffers SMALLER than that -- received, truncated to their size --
 * use the *_bounded entry points below, which read nothing beyond the bytes they are told exist. */
/* ansDecodeBatchStride, GpuANSCodec.h:170-226 / GpuANSDecode.cu:20-45 */
int dgpu_ans_decode_batch_stride(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* in_dev, uint32_t inPerBatchStride,
    void* out_dev, uint32_t outPerBatchStride, uint32_t outPerBatchCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev,
    void* stream, int32_t* errBatch);

/* ansDecodeBatchPointer, GpuANSCodec.h:228-263 / GpuANSDecode.cu:47-120 */
int dgpu_ans_decode_batch_pointer(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in,
    void* const* out, const uint32_t* outCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev,
    void* stream, int32_t* errBatch);

/* ansDecodeBatchSplitSize, GpuANSCodec.h:265-304 / GpuANSDecode.cu:122-193 */
int dgpu_ans_decode_batch_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in,
    void* out_dev, const uint32_t* outSplitSizes,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev,
    void* stream, int32_t* errBatch);

/* ---- rANS info ----------------------------------------------------------- */
/* ansGetCompressedInfo, GpuANSCodec.h:310-324 / GpuANSInfo.cu:13-35 (host
 * array of device pointers) */
int dgpu_ans_get_compressed_info(
    void* temp_dev, size_t tempBytes,
    const void* const* in, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outChecksum_dev, void* stream);
/* ansGetCompressedInfoDevice, GpuANSCodec.h:326-341 / GpuANSInfo.cu:37-49 */
int dgpu_ans_get_compressed_info_device(
    const void* const* in_dev, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outChecksum_dev, void* stream);

/* ---- float codec --------------------------------------------------------- */
/* floatCompress, GpuFloatCodec.h:103-141 / GpuFloatCompress.cu:47-101 */
int dgpu_float_compress(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in, const uint32_t* inSize,
    void* const* out,
    uint32_t* outSize_dev,
    void* stream);

/* floatCompressSplitSize, GpuFloatCodec.h:143-178 / GpuFloatCompress.cu:103-159 */
int dgpu_float_compress_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* in_dev, const uint32_t* inSplitSizes,
    void* out_dev, uint32_t outStride,
    uint32_t* outSize_dev,
    void* stream);

/* floatDecompress, GpuFloatCodec.h:184-218 / GpuFloatDecompress.cu:22-115 */
int dgpu_float_decompress(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in,
    void* const* out, const uint32_t* outCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev,
    void* stream, int32_t* errBatch);

/* floatDecompressSplitSize, GpuFloatCodec.h:220-258 / GpuFloatDecompress.cu:117-179 */
int dgpu_float_decompress_split_size(
    void* temp_dev, size_t tempBytes, size_t* tempUsed,
    uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch,
    const void* const* in,
    void* out_dev, const uint32_t* outSplitSizes,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev,
    void* stream, int32_t* errBatch);

/* ---- decode with known input sizes (no upstream equivalent) -------------------
 * The reference's decode API carries no compressed sizes (GpuANSCodec.h:228-304,
 * GpuFloatCodec.h:184-258): an archive that was cut short is followed past its buffer.
 * These variants take `inBytes` (HOST array, bytes available at in[i]); an archive whose
 * header claims more is reported through outSuccess and not read.  The tensor-level
 * ops (decompress_data*, DietGpu.cpp:530-911), which know every input tensor's size,
 * go through them. */
int dgpu_ans_decode_batch_pointer_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes,
    void* const* out, const uint32_t* outCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch);
int dgpu_ans_decode_batch_split_size_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes,
    void* out_dev, const uint32_t* outSplitSizes,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch);
int dgpu_float_decompress_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes,
    void* const* out, const uint32_t* outCapacity,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch);
int dgpu_float_decompress_split_size_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* const* in, const uint32_t* inBytes,
    void* out_dev, const uint32_t* outSplitSizes,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch);

/* ---- float stride batches with capacities (no upstream equivalent) ---------------
 * For exchanging compressed rows at a FIXED width (README.md:68-72,104: compressed collectives): the rows of one
 * tensor are compressed straight into the rows of the send matrix, `outStrideBytes` apart, and nothing is stored
 * beyond `outCapacityBytes` of a row -- a row whose archive is longer (outSize_dev[i] > outCapacityBytes) is
 * incomplete and must be sent another way; outSize_dev reports the full size either way.  The capacity must hold
 * everything but the block data (header, tables, non-compressed planes: DGPU_ERR_INVALID_ARGUMENT otherwise), rows
 * have more than 4096 words.  The receiving side decodes rows of `inBytes` available bytes each; a row whose
 * archive claims more is reported through outSuccess_dev and not read (as the *_bounded entry points above). */
int dgpu_float_compress_stride_capped(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inWords, uint32_t inStrideBytes,
    void* out_dev, uint32_t outStrideBytes, uint32_t outCapacityBytes, uint32_t* outSize_dev, void* stream);
int dgpu_float_decompress_stride_bounded(
    void* temp_dev, size_t tempBytes, size_t* tempUsed, uint32_t floatType, int probBits, int useChecksum,
    uint32_t numInBatch, const void* in_dev, uint32_t inStrideBytes, uint32_t inBytes,
    void* out_dev, uint32_t outStrideBytes, uint32_t outCapacityWords,
    uint8_t* outSuccess_dev, uint32_t* outSize_dev, void* stream, int32_t* errBatch);

/* floatGetCompressedInfo, GpuFloatCodec.h:264-277 / GpuFloatInfo.cu:17-45 */
int dgpu_float_get_compressed_info(
    void* temp_dev, size_t tempBytes,
    const void* const* in, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outTypes_dev, uint32_t* outChecksum_dev,
    void* stream);
/* floatGetCompressedInfoDevice, GpuFloatCodec.h:279-292 / GpuFloatInfo.cu:47-64 */
int dgpu_float_get_compressed_info_device(
    const void* const* in_dev, uint32_t numInBatch,
    uint32_t* outSizes_dev, uint32_t* outTypes_dev, uint32_t* outChecksum_dev,
    void* stream);

/* ---- building blocks exposed for parity tests ----------------------------- */
/* ansHistogramBatch, GpuANSStatistics.cuh:384-412: [B][256] u32 counts of a
 * strided batch (any byte alignment). */
int dgpu_ans_histogram_batch_stride(
    uint32_t numInBatch,
    const void* in_dev, uint32_t inPerBatchSize, uint32_t inPerBatchStride,
    uint32_t* histogram_dev,
    void* stream);
/* ansCalcWeights, GpuANSStatistics.cuh:414-430: normalised {pdf,cdf,magic,shift}
 * uint4[B][256] in the REFERENCE's table layout (the encoder itself uses a
 * re-packed table). */
int dgpu_ans_calc_weights(
    uint32_t numInBatch, int probBits,
    const uint32_t* sizes_dev, uint32_t uniformSize,
    const uint32_t* histogram_dev,
    uint32_t* table_dev,
    void* stream);

/* ---- library-owned device state (no upstream equivalent) -------------------- */
/* The library keeps a little device memory per (device, stream) it has been used
 * on: the temp-memory overflow slab (what StackDeviceMemory.cpp:119-139 would
 * cudaMalloc/cudaFree per call) and the zero-at-rest hand-off counters of the
 * histogram pass.  It is bounded (the least recently used idle states are dropped
 * beyond 32 streams), and can be given back explicitly: synchronises the device,
 * frees the memory kept for `stream` on the current device (or for every stream),
 * returns the number of states released.  Call before destroying a stream. */
int dgpu_release_stream_state(void* stream);
int dgpu_release_all_stream_state(void);
/* Test hook: number of (device, stream) states currently kept. */
uint32_t dgpu_debug_stream_state_count(void);

/* ---- instrumentation (no upstream equivalent) ------------------------------ */
/* Per-kernel timing with HIP events recorded on the launch stream; used by
 * bench.py for the roofline figure.  Off by default. */
void dgpu_prof_enable(int on);
void dgpu_prof_reset(void);
/* Writes a JSON object {"kernel": {"launches": n, "total_ms": t}, ...} into buf
 * (synchronises on the recorded events). Returns length, or -1 if cap is too small. */
int dgpu_prof_summary(char* buf, size_t cap);

/* Test hook: makes every encoder workgroup whose index is congruent to 1 modulo
 * `modulo` start about half a millisecond after the others (0 = off, the
 * default).  It emulates a grid that is only partly resident at first -- other
 * kernels holding compute units -- so that tests can exercise the path in which
 * running workgroups take over the tiles of workgroups that have not started. */
void dgpu_debug_set_absent_workgroups(uint32_t modulo);

/* Measurement / test hook: how the workgroups of the tiled encoder (k_ans_encode; replaces the grid of ansEncodeBatch,
 * GpuANSEncode.cuh:429-461) come to their tiles -- for raw bytes and for float tiles of 2 or 4 blocks, the kernels that
 * exist in both forms.  -1 (default): the library decides per call (one workgroup per tile when there are more tiles
 * than resident workgroups); 0: as many persistent workgroups as fit on the device, static ticket map; 1: one workgroup
 * per tile, dispatched by the hardware in ticket order.  Archives are byte-identical either way.  8-block float tiles
 * always run persistent, single-block elements always one workgroup per pair. */
void dgpu_debug_set_encoder_dispatch(int mode);

/* Measurement / test hook: the order in which the workgroups of the tiled decoder (k_ans_decode; replaces the grid of
 * ansDecodeBatch, GpuANSDecode.cuh:299-403) take the (element, tile) pairs.  -1 (default): the library decides per
 * call; 0: element-major; 1: tile-major; 2: every XCD walks its own elements.  Outputs are identical either way. */
void dgpu_debug_set_decoder_order(int order);

/* Measurement / test hook: batches whose elements differ widely in size (the tensors of a model in one call).  The
 * grids of upstream's kernels -- and this library's -- are rectangles laid out for the largest element
 * (GpuANSEncode.cuh:692-760, GpuANSDecode.cuh:299-403: maxSize x numInBatch); here the sizes arrive as host arrays, so
 * for a batch in which at least a fifth of that rectangle would be empty the host lists the tiles and histogram parts
 * that exist and the kernels work through the lists.  -1 (default): that policy; 0: always the rectangles; 1: the
 * lists for every pointer-array call whose sizes differ.  Archives and outputs are byte-identical either way. */
void dgpu_debug_set_work_lists(int mode);

/* Measurement / test hook: SIZE CLASSES inside one batch.  Upstream sizes the one grid of a call for its largest member
 * (GpuANSEncode.cuh:753-771, GpuANSDecode.cuh:299-403); here a pointer-array call whose members fall into more than one
 * of the classes {one block, <= 2, <= 4 (decode: <= 8), more} launches every class on the kernels of its own geometry,
 * one class after the other on the caller's stream.  -1 (default): when the smaller classes together hold at least 256
 * members, classes of fewer than 32 joining the next larger one; 0: one geometry per call; 1: every class that has a
 * member.  Needs the work lists (dgpu_debug_set_work_lists != 0).  Archives and outputs are byte-identical either way. */
void dgpu_debug_set_size_classes(int mode);

/* Measurement hook: 0 makes every pointer-array call upload its parameter block
 * (no reuse of blocks already resident on the device); 1 (default) restores the
 * cache.  bench.py uses it to report the step time without the cache next to the
 * steady-state one. */
void dgpu_debug_set_param_cache(int on);

/* Cache policy of the encoder's histogram pass (its first kernel; the second reads the same input again): 0 = the
 * input is read with non-temporal loads (default), 1 = with ordinary loads, which allocate in the 256 MiB memory-side
 * cache -- the pass then also pays for evicting whatever dirty lines sit there and the encode kernel reads its input
 * from the cache.  Worth ~9 % where the codec's own output is what fills the cache (compress followed at once by
 * decompress on the same device, on buffers that change every call), a loss of 4-20 % otherwise (DESIGN.md
 * section 5).  -1 restores the default.  Process-wide; archives are identical either way. */
void dgpu_set_histogram_load_policy(int mode);

/* HIP graphs.  A call made while its stream is being captured (hipStreamBeginCapture) bakes library-owned device
 * addresses into the graph: the resident copy of its pointer / size arrays and the stream's hand-off counters and
 * overflow slab.  The library pins them -- they are neither evicted nor trimmed nor released by
 * dgpu_release_stream_state / dgpu_release_all_stream_state -- until the caller, having destroyed every such graph,
 * calls dgpu_release_graph_state() (returns the number of objects unpinned / released; synchronises the devices).
 * Nothing can be uploaded or allocated under capture: a call whose arrays are not resident yet, or whose temp
 * memory overflows into a slab that would have to grow, fails with DGPU_ERR_HIP and a message that says so; running
 * the same call once before capturing makes everything resident. */
int dgpu_release_graph_state(void);

#ifdef __cplusplus
}
#endif
#endif /* DIETGPU_AMD_H */
