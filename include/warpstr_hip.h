/*
 * warpstr_hip.h -- C ABI of the MI355X-native WarpSTR caller (step 3: TR calling).
 *
 * The upstream project is pure Python and has no FFI for this path; its seams are Python
 * callables.  Each entry point below states the upstream interface it stands in for
 * (paths relative to the upstream repository root):
 *
 *   wsx_call_batch, wsx_call_batch_reads
 *                        <->  CallerWrapper.run(workload) -> List[CallerResult]
 *                             src/caller/wrapper.py:104-120 (Pool.map of warpstr_call_parallel,
 *                             251-289; WarpSTR.run, src/caller/caller.py:117-149)
 *   wsx_warp_batch       <->  WarpSTR.warp(signal, mask) -> WarpResult(trace)
 *                             src/caller/caller.py:189-193 (_calc_dtw_astates 198-245,
 *                             _backtracking 247-301)
 *   wsx_prepare_signals  <->  Fast5.get_data_processed for every `saved` read (get_workload),
 *                             src/schemas/fast5.py:45-57, src/caller/wrapper.py:44-54
 *   wsx_vbz_decode       <->  what h5py hands `Fast5.get_data_processed` when it reads `Raw/Signal` (src/schemas/fast5.py:50-52):
 *                             the samples of a VBZ-filtered dataset.  Upstream leaves the decoding to the HDF5 filter plugin
 *                             (filter id 32020, ont-vbz-hdf-plugin: absent from the upstream tree); this entry point does
 *                             the part of it behind the zstd frame -- StreamVByte, zig-zag, running sum -- on the device
 *   wsx_locate_flanks    <->  find_sequence(text, pattern) for a batch of (basecalled window, flank) pairs,
 *                             src/extractor/tr_extractor.py:196-250 (align_seq 253-274 calls it twice per read);
 *                             the alignment itself is Bio.pairwise2.align.localms (biopython ==1.75, absent here:
 *                             parity of this entry point is UNPINNED, see oracle/flank_oracle.c)
 *   wsx_moves_to_raw     <->  transform_moves + extract_from_moves, tr_extractor.py:147-193
 *   wsx_caller_create    <->  CallerWrapper.__init__ / init_pool: automata + config made
 *                             available to the workers, src/caller/wrapper.py:63-70,92-102
 *   wsx_automaton        <->  StateAutomata(states, endstate, mask), src/caller/automata.py:36-48
 *   wsx_params           <->  caller_config / rescaler_config, src/config.py:91-119
 *   wsx_result           <->  CallerResult, src/caller/caller.py:46-51 (lengths instead of
 *                             strings: only len(seq)/len(resc_seq) reach overview.csv,
 *                             src/caller/overview.py:57-73; strings are rebuilt from the traces)
 *
 * Conventions
 *   - Plain pointers and sizes only.  `mem` says where the bulk buffers live
 *     (WSX_MEM_HOST: ordinary host memory, the library stages it; WSX_MEM_DEVICE: HBM of the
 *     caller's device, e.g. a torch tensor's data_ptr()).  Metadata arrays (offsets, automaton
 *     ids) are always host memory.
 *   - All buffers are caller-owned; the library keeps no reference after a call returns.
 *   - Reads are independent.  A failure of one read (the upstream code would raise and abort the
 *     whole Pool.map, src/caller/caller.py:290-291, 395-397) is reported in wsx_result.status and
 *     the other reads are unaffected.  The function return value reports process-level errors.
 *   - Results are positionally aligned with the input order (as Pool.map is).
 *   - One wsx_caller per device and host thread; calls on one handle are serialised by the caller.
 */
#ifndef WARPSTR_HIP_H
#define WARPSTR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WSX_ABI_VERSION 14

/* function return codes */
enum {
    WSX_SUCCESS = 0,
    WSX_ERR_INVALID = -1,     /* bad argument */
    WSX_ERR_NO_DEVICE = -2,   /* no usable HIP device */
    WSX_ERR_HIP = -3,         /* a HIP runtime call failed; see wsx_last_error() */
    WSX_ERR_UNSUPPORTED = -4, /* configuration outside what the kernels implement */
    WSX_ERR_NOMEM = -5,
};

/* per-read status (wsx_result.status); 0 = called */
enum {
    WSX_READ_OK = 0,
    WSX_READ_SHAPE = 1,         /* T <= min_values_per_state (upstream: IndexError, caller.py:206-208) */
    WSX_READ_BACKTRACK = 2,     /* upstream RuntimeError, caller.py:290-291 */
    WSX_READ_FIT_POINTS = 3,    /* < 4 states pass filter_alignment (upstream: splrep TypeError) */
    WSX_READ_FIT_ORDER = 4,     /* degenerate abscissae for the rescaling fit */
    WSX_READ_FIT_SMOOTH = 5,    /* FITPACK's smoothing spline (threshold > 1 only) has a coefficient that is not finite */
    WSX_READ_NO_REPEAT = 6,     /* no repeat state on the path (upstream IndexError, caller.py:384) */
    WSX_READ_SEGMENT_RANGE = 7, /* upstream IndexError in find_event_borders/segment (caller.py:395-397) */
};

enum { WSX_MEM_HOST = 0, WSX_MEM_DEVICE = 1 };

/* One k-mer state automaton (host pointers; copied to the device by wsx_caller_create). */
typedef struct wsx_automaton {
    int32_t n_states;           /* S */
    int32_t endstate;           /* StateAutomata.endstate */
    int32_t flank_length;       /* Locus.flank_length used to build it */
    int32_t reverse;            /* 1: sequences of this automaton are reverse-complemented (reverse-strand reads,
                                   WarpSTR._get_sequence, src/caller/caller.py:187) */
    const double *value;        /* [S] State.value: expected normalised level */
    const int32_t *seq_idx;     /* [S] State.seq_idx */
    const int32_t *pred_ptr;    /* [S+1] CSR offsets of State.incoming */
    const int32_t *pred_idx;    /* [pred_ptr[S]] predecessor state ids, in `incoming` order */
    const uint8_t *repeat_mask; /* [S] StateAutomata.mask */
    const uint8_t *last_base;   /* [S] ASCII of State.kmer[-1]; may be NULL (then no sequences can be requested) */
} wsx_automaton;

typedef struct wsx_params {
    int32_t min_values_per_state; /* tr_calling_config.min_values_per_state, default 4 (> 1) */
    int32_t states_in_segment;    /* tr_calling_config.states_in_segment, default 6 (> 1) */
    double threshold;             /* rescaling.threshold, default 0.5 (> 0; above 1 FITPACK's smoothing branch can be taken) */
    double max_std;               /* rescaling.max_std, default 0.5 */
    int32_t method_median;        /* rescaling.method: 0 = mean, 1 = median */
    int32_t reps_as_one;          /* rescaling.reps_as_one (0/1) */
} wsx_params;

typedef struct wsx_result {
    int32_t status;       /* WSX_READ_* */
    int32_t len1;         /* len(CallerResult.seq)      -> overview.csv `orig` */
    int32_t len2;         /* len(CallerResult.resc_seq) -> overview.csv `results` (allele length) */
    int32_t n_trans1;     /* states visited by the first alignment */
    int32_t n_trans2;     /* states visited by the second alignment */
    int32_t reserved;
    double cost1;         /* CallerResult.cost      -> `dtw_cost1` */
    double cost2;         /* CallerResult.resc_cost -> `dtw_cost2` */
    double dtw_end_cost1; /* D[T-1, endstate] of the first DP */
    double dtw_end_cost2; /* D[T-1, endstate] of the second DP */
} wsx_result;

/* Optional per-sample outputs of wsx_call_batch (any pointer may be NULL). Same `mem` as the
 * signal; laid out with the same offsets[] as the signal. */
typedef struct wsx_traces {
    uint16_t *trace1;    /* state id per sample, first alignment */
    uint16_t *trace2;    /* state id per sample, alignment of the rescaled signal */
    double *rescaled;    /* rescale_signal(signal, alignment), caller.py:124 */
    uint8_t *badmask;    /* mask_bad_repeats(...)[2], caller.py:125-126 */
    uint8_t *seq1;       /* CallerResult.seq as ASCII: read r occupies [offsets[r], offsets[r] + len1) */
    uint8_t *seq2;       /* CallerResult.resc_seq as ASCII: [offsets[r], offsets[r] + len2) */
} wsx_traces;

typedef struct wsx_caller wsx_caller;

/* Library / device queries. */
int wsx_abi_version(void);
int wsx_device_count(void);
const char *wsx_last_error(void);

/*
 * Create a caller bound to HIP device `device` holding `n_automata` automata (for one locus:
 * 0 = template strand, 1 = reverse strand; more for mixed-locus batches).
 * `stream` is a hipStream_t passed as void* (NULL = the device's default stream); all work of
 * this handle is enqueued on it.
 */
int wsx_caller_create(wsx_caller **out, int device, const wsx_automaton *automata, int32_t n_automata,
                      const wsx_params *params, void *stream);
void wsx_caller_destroy(wsx_caller *c);

/*
 * Append `n_automata` automata to a handle that is in use: *first_index receives the index of the first of them (the
 * `automaton_id` the reads of later calls name them by; earlier automata keep theirs).  Upstream builds the two automata of a
 * locus when its loop reaches that locus (WarpSTR.py:33-76 -> CallerWrapper.__init__, src/caller/wrapper.py:63-70); a handle for
 * all loci of a run would otherwise have to see every locus before the first read is called -- this way the loci of group g+1 are
 * parsed, compiled, placed and uploaded while the reads of group g are on the device.
 * May be called while earlier calls of the handle are still running on the device (they keep the table they were enqueued
 * with); must not be called concurrently with another entry point of the SAME handle.  On failure the handle is unchanged.
 */
int wsx_caller_add_automata(wsx_caller *c, const wsx_automaton *automata, int32_t n_automata, int32_t *first_index);

/*
 * Upper bound, in bytes, of the device workspace the handle may allocate.  Default: 60 % of the device memory that is free
 * when the handle is created (at least 2 GiB) -- the handle allocates what a call needs, the limit only decides when a call
 * is cut into more chunks than its size asks for.  (Upstream has no counterpart: a Pool worker holds one T x S matrix.)
 */
int wsx_caller_set_workspace_limit(wsx_caller *c, uint64_t bytes);
int wsx_caller_get_workspace_limit(wsx_caller *c, uint64_t *bytes);

/*
 * Launch-policy knobs of a handle; the defaults are the measured ones.  Tests use them to force the
 * fallback kernels; upstream's counterpart is the `threads` argument of CallerWrapper (src/caller/wrapper.py:63-70,104-109):
 * how the work is spread, never what is computed.
 */
enum {
    WSX_TUNE_STREAM_TRACEBACK_MIN = 1, /* smallest single-slot launch (reads) that takes the thread-per-read traceback; 8192 */
    WSX_TUNE_BORDERS_WAVE_BELOW = 2,   /* launches of fewer reads take the wave-per-read borders stage; 8192 */
    WSX_TUNE_SEGMENT_TWO_KERNELS = 3,  /* 1: always the two-kernel segmentation (otherwise only for reads beyond ~90 k samples); 0 */
    WSX_TUNE_FILL_BLOCKS_PER_CU = 4,   /* > 0: cap the fill's workgroups per CU; 0 */
    WSX_TUNE_CHUNKS = 5,               /* > 0: chunks per call instead of the built-in rule (also WSX_CHUNKS at creation); 0 */
    WSX_TUNE_SMALL_PIPE_SAMPLES = 6,   /* pipelined calls up to this many samples stay in one chunk; 52 Mi */
    WSX_TUNE_CALLS_IN_FLIGHT = 7,      /* pipelined calls the host may run ahead of the device, 2..4; 2 */
    WSX_TUNE_SMALL_CALLS_IN_FLIGHT = 8 /* ... for one-chunk calls, 2..4; 4 */
};
int wsx_caller_set_tuning(wsx_caller *c, int32_t knob, int64_t value);

/*
 * Number of HIP streams (and workspace sets) the handle may use (1..8, default 4; 1 = everything on the handle's
 * stream).  A big call spreads its chunks over four of them; small pipelined calls take two each and alternate, so that
 * consecutive calls run side by side.
 */
int wsx_caller_set_streams(wsx_caller *c, int32_t n_streams);

/*
 * Pipelined calls (device buffers only; host-buffer calls stay synchronous).  Upstream calls one locus after the
 * other (WarpSTR.py:66-76, main_wrapper per locus); a caller that has the next batch ready does not want the GPU to
 * drain in between.  With `on` != 0 a wsx_call_batch / wsx_warp_batch on WSX_MEM_DEVICE buffers still reads its inputs
 * in the order of the handle's stream, but no longer makes that stream wait for its end: the next call's chunks then
 * follow this call's on every internal stream without a gap (at most two calls are in flight, four if they are small --
 * fewer than 32 768 reads and at most 52 M samples --; the next one blocks the host until the oldest has finished: keep that many
 * sets of output buffers).  Outputs may be consumed only after wsx_caller_join (stream order) or
 * wsx_caller_synchronize / wsx_caller_last_timing (host).  Turning the mode off joins the handle's stream.
 * Handles created with rescaling.threshold > 1 synchronise the host once per chunk inside every call (the number of reads
 * that take FITPACK's smoothing branch sizes its workspace), so their calls overlap less.
 */
int wsx_caller_set_pipelined(wsx_caller *c, int32_t on);

/*
 * Make `stream` (a hipStream_t; NULL = the handle's stream) wait for every call enqueued on this handle so far.
 * Does not block the host.  A no-op outside pipelined mode, where calls already end on the handle's stream.
 */
int wsx_caller_join(wsx_caller *c, void *stream);

/*
 * Call a batch of reads: both alignments, rescaling and bad-repeat masking, per read.
 *   signal       concatenated normalised squiggles, float64 (ReadSignal.signal), in `mem`
 *   offsets      host int64[n_reads+1]; read r is signal[offsets[r] .. offsets[r+1])
 *   automaton_id host int32[n_reads]; which automaton each read is aligned to
 *                (upstream: rev_sta if ReadSignal.reverse else temp_sta, wrapper.py:279-286)
 *   results      wsx_result[n_reads] in `mem`
 *   traces       optional extra outputs (NULL for none)
 * Synchronous for WSX_MEM_HOST.  For WSX_MEM_DEVICE the work is enqueued on the handle's stream
 * and the call returns without waiting; use wsx_caller_synchronize (or the stream) before reading.
 */
int wsx_call_batch(wsx_caller *c, int mem, const double *signal, const int64_t *offsets,
                   const int32_t *automaton_id, int64_t n_reads, wsx_result *results, const wsx_traces *traces);

/*
 * The same for reads that live in separate host arrays -- the reference's workload is a list of ReadSignal objects, each
 * with its own numpy array (src/schemas/readsignal.py:6-10; get_workload, src/caller/wrapper.py:44-54): reads[r] points at
 * lengths[r] float64 samples.  The library gathers them into its pinned upload buffers with its copy threads while earlier
 * pieces are already on their way to the device, so the caller need not build one 8-bytes-per-sample buffer first.  Host
 * memory only; per-sample outputs in `traces` are laid out back to back by the running sum of `lengths`.  Synchronous.
 */
int wsx_call_batch_reads(wsx_caller *c, const double *const *reads, const int64_t *lengths, const int32_t *automaton_id,
                         int64_t n_reads, wsx_result *results, const wsx_traces *traces);

/*
 * One DP + traceback per read (WarpSTR.warp).
 *   mask         optional uint8 per sample (same offsets; NULL = unmasked), in `mem`
 *   trace        uint16 per sample, in `mem`
 *   end_cost     optional float64[n_reads]: D[T-1, endstate], in `mem`
 *   last_row     optional float64, automaton_row_stride doubles per read: D[T-1, :], in `mem`
 *   status       optional int32[n_reads] (WSX_READ_*), in `mem`
 */
int wsx_warp_batch(wsx_caller *c, int mem, const double *signal, const int64_t *offsets,
                   const int32_t *automaton_id, int64_t n_reads, const uint8_t *mask, uint16_t *trace,
                   double *end_cost, double *last_row, int32_t last_row_stride, int32_t *status);

/*
 * Raw squiggles -> normalised STR segments (Fast5.get_data_processed, src/schemas/fast5.py:45-57; the per-read part of
 * get_workload, src/caller/wrapper.py:44-54): spike removal in the raw int16 domain (brute_remove, fast5.py:90-101),
 * MAD normalisation over the WHOLE read (normalize_signal_mad, 104-114), then the slice
 * [seg_start[r] : seg_end[r] + 1] (overview.csv columns l_start_raw, r_end_raw).
 *   raw            concatenated int16 raw signals of whole reads, in `mem`
 *   raw_offsets    host int64[n_reads+1]
 *   seg_start/end  host int64[n_reads] (inclusive end, as upstream)
 *   spike_removal  0 = None, 1 = Brute, 2 = median3, 3 = median5 (tr_calling_config.spike_removal; remove_spikes,
 *                  src/schemas/fast5.py:68-75; the median filters are scipy.signal.medfilt: zero padding at both ends)
 *   signal_out     float64 output, concatenated by out_offsets (host int64[n_reads+1], lengths must equal the slice
 *                  lengths); can be passed straight to wsx_call_batch with the same offsets
 *   shift_scale    optional float64[2*n_reads] (shift, scale per read), in `mem`
 * Synchronous for WSX_MEM_HOST.  For WSX_MEM_DEVICE the work is enqueued on the handle's stream and the call returns
 * without waiting (the host metadata arrays are copied before it returns); a wsx_call_batch on the same handle that
 * follows reads signal_out in stream order.
 */
int wsx_prepare_signals(wsx_caller *c, int mem, const int16_t *raw, const int64_t *raw_offsets, const int64_t *seg_start,
                        const int64_t *seg_end, int64_t n_reads, int32_t spike_removal, double *signal_out,
                        const int64_t *out_offsets, double *shift_scale);

/*
 * The samples of VBZ-compressed signal datasets, decoded on the device (the step between the file and wsx_prepare_signals).
 * A VBZ chunk (HDF5 filter 32020, version 0, 2-byte integers) is a u32 byte count and a zstd frame; inside the frame is a
 * StreamVByte block: ceil(n/4) key bytes -- two bits per value, byte length - 1, first value in the low bits -- followed by
 * the values' little-endian bytes back to back; the values are the (optionally zig-zag mapped) differences of consecutive
 * samples, the first against 0.  The host undoes zstd (a byte-serial entropy decoder); what is left is two prefix sums --
 * where a value's bytes start, and the running sum of the differences -- which this entry point computes a workgroup per block.
 *   src            device: the blocks' bytes (StreamVByte blocks as they leave zstd, or plain little-endian int16 samples)
 *   blocks         host wsx_vbz_block[n_blocks]: where a block lies in src, what it is, where its samples go in dst.
 *                  Checked before anything is enqueued (WSX_ERR_INVALID: a block outside src / dst, n_values < n_samples, fewer
 *                  bytes than its key area plus one byte per value); copied before the call returns
 *   dst            device int16: sample dst_offset + i of block b is its i-th sample (the running sum wraps as int16 does)
 *   status         device int32[n_blocks] or NULL: 0, or 1 for a block whose keys ask for more bytes than it has (its
 *                  samples from there on are those of zero bytes; nothing outside the block is read).  Valid in stream order
 * Enqueued on the handle's stream; returns without waiting.  A wsx_prepare_signals on the same handle that follows reads dst in
 * stream order.
 */
typedef struct wsx_vbz_block {
    int64_t src_offset; /* first byte of the block in src */
    int64_t src_bytes;  /* its size */
    int64_t dst_offset; /* its first sample in dst (in samples) */
    int32_t n_samples;  /* samples wanted of it: its first n_samples */
    int32_t kind;       /* WSX_VBZ_PLAIN: int16 samples; WSX_VBZ_SVB_ZIGZAG / WSX_VBZ_SVB: StreamVByte of (zig-zag) differences */
    int32_t n_values;   /* values the block codes (>= n_samples; its key area is ceil(n_values / 4) bytes): HDF5 hands a filter the
                           whole chunk, so the last chunk of a dataset codes chunk-length values of which the dataset holds fewer */
    int32_t reserved;   /* 0 */
} wsx_vbz_block;
enum { WSX_VBZ_PLAIN = 0, WSX_VBZ_SVB_ZIGZAG = 1, WSX_VBZ_SVB = 2 };
int wsx_vbz_decode(wsx_caller *c, const uint8_t *src, int64_t src_bytes, const wsx_vbz_block *blocks, int64_t n_blocks,
                   int16_t *dst, int64_t dst_samples, int32_t *status);

/*
 * The zstd frames of VBZ chunks, decoded on the device (the step in front of wsx_vbz_decode): what the HDF5 filter plugin 32020 asks
 * of libzstd when h5py reads `Raw/Signal` for Fast5.get_data_processed (src/schemas/fast5.py:50-52; plugin and libzstd are third-party
 * dependencies that are not in the upstream tree; the format is RFC 8878).  Four kernels: the Huffman-coded blocks of all frames are
 * listed, the list ordered (most literals first), the literals decoded four blocks to a wavefront (the four streams of a block side
 * by side), then a wavefront per frame executes its blocks' sequences in order.
 *   src        device: the frames' bytes (each from its magic number on)
 *   frames     host wsx_zstd_frame[n_frames]: where a frame lies in src, where its content goes in dst and how many bytes that is
 *              (the content size the frame's header declares).  Checked before anything is enqueued; copied before the call returns
 *   dst        device: the content of frame i at dst_offset (e.g. the StreamVByte block wsx_vbz_decode then takes from there)
 *   scratch    device, as large as dst: where the literals of a frame's blocks lie between the two phases
 *   status     device int32[n_frames] or NULL: 0 decoded; 1 the frame uses what this decoder leaves to the host (a dictionary,
 *              more than 32 blocks): decompress it there; 2 corrupt (or its
 *              content is not dst_bytes long).  Output of a frame with a non-zero status is undefined.  Valid in stream order
 * Enqueued on the handle's stream; returns without waiting.  A wsx_vbz_decode on the same handle that follows reads dst in stream
 * order.
 */
typedef struct wsx_zstd_frame {
    int64_t src_offset; /* first byte of the frame in src */
    int64_t src_bytes;  /* its size */
    int64_t dst_offset; /* where its content goes in dst (and its literals in scratch) */
    int64_t dst_bytes;  /* the content size it declares (at most 4 MB: 32 blocks) */
} wsx_zstd_frame;
int wsx_zstd_decode(wsx_caller *c, const uint8_t *src, int64_t src_bytes, const wsx_zstd_frame *frames, int64_t n_frames, uint8_t *dst,
                    int64_t dst_bytes, uint8_t *scratch, int32_t *status);

int wsx_caller_synchronize(wsx_caller *c);

/*
 * Timing of the most recent wsx_call_batch on this handle, measured with HIP events on the handle's
 * stream: total milliseconds in the DP fill kernels (both passes) and number of such launches, total
 * milliseconds of the whole enqueue..finish region.  Blocks until the work has finished.
 */
int wsx_caller_last_timing(wsx_caller *c, double *dp_kernel_ms, int32_t *dp_launches, double *total_ms);

/*
 * Device memory the handle holds (automata, metadata, every work set of every stream, loader pool), and the caller's part
 * of it (everything but the loader pool) per sample of the most recent call: the workspace is sized per sample -- the
 * read's normalised and rescaled signal, run lists, alignment records, fit pairs, scratch and the DP back-pointers
 *  -- for every chunk in flight.
 */
int wsx_caller_workspace(wsx_caller *c, uint64_t *bytes_allocated, double *bytes_per_sample);

/*
 * With `on` != 0 the figures of wsx_caller_last_timing cover every call from now on instead of the most recent one
 * (the fill kernels' events of successive calls are kept, total_ms runs from the first call's start to the last
 * call's end); calling it again, with either value, starts afresh.  For measurements over several pipelined calls
 * (helpers.print_time_duration brackets whole steps upstream, src/helpers.py:16-29).
 */
int wsx_caller_timing_window(wsx_caller *c, int32_t on);

/*
 * The fill launches behind wsx_caller_last_timing, one by one: begin/end of launch i in milliseconds since the start of
 * the timed call (or of the timing window), and the number of reads it covered.  Launches on different streams overlap,
 * so their union -- not their sum -- is the time the device spent with a fill kernel in flight.  At most `capacity`
 * entries are written (any array may be NULL); *n_out receives the number of launches.  Blocks like last_timing.
 */
int wsx_caller_fill_intervals(wsx_caller *c, double *begin_ms, double *end_ms, int32_t *reads, int32_t capacity,
                              int32_t *n_out);

/*
 * Where wsx_caller_create spent its time, in seconds (a handle for all loci of a run holds thousands of automata: placing
 * their states is host work).  seconds[0] validation and sizing, [1] state placement (wsx_place.h, on host threads),
 * [2] packing the tables, [3] upload, [4] streams and events; at most `capacity` entries are written.  No upstream counterpart.
 */
int wsx_caller_create_times(const wsx_caller *c, double *seconds, int32_t capacity);

/* Name of the DP fill kernel variant used for automaton `a` (for profiles), e.g. "dtw_fill_fast<4, 1, 2, 2>". */
const char *wsx_caller_kernel_name(wsx_caller *c, int32_t a);

/* ---- flank localisation (upstream step 1) -------------------------------------------------------- */

/* alignment_config, src/config.py:135-141 (upstream: 2, -3, -3, -3).  gap_open must equal gap_extend. */
typedef struct wsx_align_scores {
    int32_t match, mismatch, gap_open, gap_extend;
} wsx_align_scores;

/* One find_sequence result (Alignment, tr_extractor.py:45-63, before the accuracy/identity gate of __post_init__). */
typedef struct wsx_flank_hit {
    int32_t status;       /* 0 found; 1 no positive-scoring alignment (upstream: IndexError on pairwise2's empty list) */
    int32_t score;        /* pairwise2 score + find_sequence's correction (tr_extractor.py:243-245) */
    int32_t start, end;   /* Position(real_start, end) before origin_offset: text coordinates of the whole flank */
    int32_t matches;      /* identity = matches / span (tr_extractor.py:234-246) */
    int32_t span;         /* len(ref) */
    int32_t row0, col0;   /* text / pattern bases in front of the local alignment */
    int32_t row1, col1;   /* text / pattern bases up to its end */
    int32_t gaps_text;    /* '-' in the aligned text inside the local region (nums_gaps) */
    int32_t gaps_pattern; /* '-' in the aligned pattern inside the local region (nums_gaps2) */
    int32_t raw_score;    /* the alignment's own score */
    int32_t n_ops;        /* length of the local region */
    /* How much of this hit rests on a tie rule (upstream takes pairwise2's first alignment; Biopython's order among equally
     * good alignments is not restated): both 0 = the optimal local alignment is UNIQUE, and every correct Smith-Waterman --
     * pairwise2 included -- returns exactly this one. */
    int32_t n_best_cells; /* cells of the whole matrix that reach the best score (1 = the end of the alignment is unique) */
    int32_t tie_steps;    /* steps of the traceback where more than one predecessor reproduces the cell's score */
} wsx_flank_hit;

/*
 * Locates pattern r (pattern[pattern_offsets[r] .. pattern_offsets[r+1]), 1..256 bases) in text r for r = 0..n-1.
 * text / pattern / hits / ops live where `mem` says; the offset arrays are host memory.  ops (optional): per pair
 * ops_stride bytes, ops_stride >= 2 * (longest pattern) + 8, filled with the local region's operations in text order
 * ('M' base against base, 'U' text base against a gap, 'L' pattern base against a gap), zero-padded -- enough to
 * rebuild upstream's Mapping(ref, mapping, query) strings.  Blocks until the results are written.
 */
int wsx_locate_flanks(int device, void *stream, int mem, const uint8_t *text, const int64_t *text_offsets,
                      const uint8_t *pattern, const int64_t *pattern_offsets, int64_t n, const wsx_align_scores *scores,
                      wsx_flank_hit *hits, uint8_t *ops, int32_t ops_stride);

/*
 * Raw-signal positions of basecalled positions through Guppy's move table: for read r with moves
 * moves[move_offsets[r] .. move_offsets[r+1]) (one byte per block), raw_start = strand_start + block_stride * (first block
 * whose context index equals pos_start), raw_end = strand_start + block_stride * (last block whose context index equals
 * pos_end); -1 where no block has that index.  moves / raw_start / raw_end live where `mem` says; the per-read
 * parameter arrays are host memory.
 */
int wsx_moves_to_raw(int device, void *stream, int mem, const uint8_t *moves, const int64_t *move_offsets,
                     const int32_t *pos_start, const int32_t *pos_end, const int64_t *strand_start,
                     const int32_t *block_stride, int64_t n, int64_t *raw_start, int64_t *raw_end);

#ifdef __cplusplus
}
#endif
#endif /* WARPSTR_HIP_H */
