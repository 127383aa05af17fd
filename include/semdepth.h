/*
 * semdepth.h — C ABI of libsemdepth.so, the MI355X (gfx950) implementation of semantic-depth's
 * per-frame hot path.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * The reference (pablopalafox/semantic-depth) has no FFI: its operator boundary is the Python
 * duck-type FrameProcessor consumes.  Every entry point below names the reference interface
 * it replaces (file:line relative to the reference tree; "seq" = semantic_depth_cityscapes_sequence.py).
 *
 * Conventions
 *   - every function returns sd_status (0 = OK, negative = error); sd_last_error() gives the text.
 *   - all data pointers are DEVICE pointers owned by the caller unless the name ends in _host.
 *   - every compute call takes a hipStream_t (passed as void*) and is asynchronous w.r.t. the host.
 *   - a handle is bound to one device, is not thread-safe, and owns no device memory: the caller
 *     (PyTorch-ROCm in the Python host) allocates the weight and workspace arenas whose sizes
 *     sd_query_memory() reports and binds them with sd_bind_memory().
 *   - images are NHWC, row-major, channel order as the caller supplies it (the reference feeds BGR).
 */
#ifndef SEMDEPTH_H
#define SEMDEPTH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int sd_status;
enum {
    SD_OK = 0,
    SD_ERR_INVALID = -1,   /* bad argument */
    SD_ERR_HIP = -2,       /* HIP runtime error (text in sd_last_error) */
    SD_ERR_STATE = -3,     /* call order violated (e.g. forward before weights are loaded) */
    SD_ERR_NOTFOUND = -4   /* unknown weight / tensor name */
};

typedef enum { SD_ENC_VGG = 0, SD_ENC_RESNET50 = 1 } sd_encoder;      /* semantic_depth.py:721-722 --encoder */
typedef enum { SD_NET_FCN8S = 0, SD_NET_MONODEPTH = 1 } sd_net;
/* arithmetic of the conv stacks (f32 accumulate everywhere; gfx950 has no TF32):
 *   SD_PREC_F32    exact f32 MFMA;
 *   SD_PREC_BF16X2 every f32 operand split into two bf16 (hi + lo), THREE bf16 MFMA products per product (~1e-5 relative);
 *   2-product form: the activation rounded once to fp16 (one 16-bit plane: half the bytes of every tensor), the weight split into
 *                  two fp16 (22 bits), TWO fp16 MFMA products x*w_hi + x*w_lo; the only error is the 2^-12 rounding of the
 *                  activations (3e-5 .. 2e-4 on the outputs per layer, profiles/r02_precision_calibration.json);
 *   1-product form (":1"): the same one-plane fp16 activation times w_hi only: plain fp16 x fp16, adds the 2^-12 rounding of the weights;
 *   x2 form (":x"):  fp16 hi + lo ACTIVATION planes times w_hi: TWO products x_hi*w_hi + x_lo*w_hi, the error is the weight rounding
 *                  alone (for layers whose input tensor is precision-critical; direct 3x3 layers fed by direct 3x3 layers);
 *   SD_PREC_MIXED  FCN-8s as SD_PREC_BF16X2, every monodepth layer in the 2-product form;
 *   SD_PREC_PLAN   per-layer choice between these forms: the built-in plan (sd_default_plan) was calibrated on the MI355X against the
 *                  exact-f32 engine under an error budget; sd_create_with_plan takes any other choice
 *   SD_PREC_BF16X3 fp32-grade on the bf16 MFMA: every f32 operand (activations and weights) is carried as THREE bf16 planes whose sum is
 *                  the f32 value EXACTLY (8 + 8 + 8 significand bits, f32's exponent range); a product is SIX bf16 MFMA products
 *                  (hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi, each exact in the f32 accumulator); the dropped terms are below
 *                  2^-23 of the product, the size of the rounding an f32 FMA chain commits per accumulation.  Ceiling 2500 / 6 = 417
 *                  TFLOP/s of algorithmic work against 157.3 for the f32 MFMA;
 *   SD_PREC_F16X2  fp32-grade on THREE fp16 MFMA products (round 5): an activation is two fp16 planes, hi = RNE(v) and lo = RNE((v - hi) * 2^11)
 *                  -- the scaled residual stays in fp16's normal range wherever hi does, so v is carried to 22 significand bits for every
 *                  |v| in [1.2e-4, 65504] without any calibration (4 bytes per element); a weight is two fp16 planes of w * 2^k, the power of two k
 *                  chosen PER LAYER by sd_load_weight from the tensor it is given (largest stored |w'| in [2^12, 2^13): any finite f32
 *                  weight tensor loads); a product is x_hi*w_hi + x_hi*w_lo + x_lo*(w_hi * 2^-11) (the third weight operand formed in
 *                  registers), f32 accumulate, the accumulator times 2^-k in the epilogue.  Dropped: x_lo*w_lo and the representation error of each operand, 2^-23 ..
 *                  2^-24 of the product (an f32 FMA chain commits 2^-24 of the ACCUMULATOR per step).  Activations beyond +-65504 are clamped
 *                  and counted (sd_saturation_count; the host-side classes raise on a non-zero count).  Ceiling 2500 / 3 = 833 TFLOP/s of algorithmic work. */
typedef enum { SD_PREC_F32 = 0, SD_PREC_BF16X2 = 1, SD_PREC_MIXED = 2, SD_PREC_PLAN = 3, SD_PREC_BF16X3 = 4, SD_PREC_F16X2 = 5 } sd_precision;

typedef struct sd_handle sd_handle;

/* camera of DepthFrame.__init__, semantic_depth.py:592-607 (seq:500-508), plus the disparity
 * multiplier of semantic_depth.py:109,145 (seq:105,146).  Doubles: the library rounds the Q-matrix
 * entries to float32 exactly like np.float32([...]) at semantic_depth.py:691-694. */
typedef struct {
    double cx, cy, f, b, disp_mult;
} sd_camera;

/* every literal of the reference's road-width call sites (semantic_depth.py:206-259) */
typedef struct {
    double depth;         /* --depth, :754-756 (10.0) */
    double z_cut;         /* remove_from_to(..., 2, 0.0, 7.0)        :206 */
    double mad_y;         /* remove_noise_by_mad(..., 1, 15.0)       :209 */
    double mad_x;         /* remove_noise_by_mad(..., 0, 2.0)        :212 */
    double plane_thr;     /* remove_noise_by_fitting_plane(axis=1, threshold=5.0) :215-219 */
    int32_t sor_k;        /* statistical_outlier_removal nb_neighbors=10   :234-235 */
    double sor_ratio;     /*                              std_ratio=0.5 */
    int32_t ror_n;        /* radius_outlier_removal nb_points=80           :238-239 */
    double ror_r;         /*                         radius=0.5 */
    double window;        /* +-0.05 depth window, pcl.py:283 */
    double depth_offset;  /* depth-0.02, :254-255 */
    int32_t use_o3d;      /* 0 skips the two Open3D filters */
} sd_rw_params;

/* per-frame record: what the reference prints/draws (semantic_depth.py:259; seq:232-238) plus the
 * kept-point count after every stage.  This is the record the multi-GPU driver all-gathers. */
typedef struct {
    double width;          /* |x_left - x_right|, NaN when !found */
    float x_left, x_right; /* x of the first min-x / max-x point in the depth window */
    float left_pt[3], right_pt[3];
    int32_t found;         /* 0: no road point in the window ((None,None) of pcl.py:303-304) */
    int32_t n_road;        /* projected road points (points3D[road_mask]) */
    int32_t n_zcut, n_mad_y, n_mad_x, n_plane, n_sor, n_ror;
    double plane[4];       /* Cx,Cy,Cz,C of pcl.py:168 */
} sd_rw_result;

/* ---------------------------------------------------------------- lifecycle */
const char* sd_version(void);
const char* sd_status_string(sd_status s);
/* replaces DepthFrame.__init__ + SegmentFrame.__init__ (semantic_depth.py:464-469, :575-624):
 * fixes H, W, the largest batch a call may carry and the monodepth encoder; builds both layer plans.
 * Every kernel choice that changes the order of a sum is made here, on a full network pass of the handle (sd_pass_frames), not on the frames of a
 * call: on one handle a frame's outputs are the same bits whether it is submitted alone, with others, or at another batch position.
 * Environment read HERE and nowhere later: SEMDEPTH_DISABLE=name[,name...] (dma dma3 direct stem fold tail1 pool_fuse planar n16 fuse1 fuse4 flat rowskip
 * dma_big mfma16: the generic kernel instead of the named specialised one -- parity tests and A/B runs; an unknown name fails with SD_ERR_INVALID),
 * SEMDEPTH_CHUNK (frames per network pass, default 32), SEMDEPTH_RESERVE_CUS (sd_set_reserved_cus), SEMDEPTH_PROFILE_VERBOSE=1 (per-layer sd_profile labels),
 * SEMDEPTH_KEEP_ACTIVATIONS (no arena reuse: every intermediate tensor stays readable by sd_net_tensor). */
sd_status sd_create(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, sd_precision prec);
/* the same with an explicit precision plan for the split engine: per network a comma-separated list of conv layer names
 * (sd_net_tensor names, e.g. "fc6,fc7" / "enc/res4*,dec/upconv6"; a trailing '*' matches a prefix, "*" = all, "" = none) that run
 * the 2-product fp16 scheme -- or, with the suffix ":1" / ":x", the 1-product / x2 form; the rest run the 3-product bf16 one.  The
 * choice is closed under "one plane format per tensor" (sd_precision_plan returns what actually runs, with the suffixes).  ":x" on a
 * layer the geometry does not route to the direct 3x3 kernel (or whose producer is not one) keeps three products; on a layer that
 * is no 3x3 stride-1 convolution with a multiple of 64 output channels it is SD_ERR_INVALID, as is an unknown layer name.  A per-pixel
 * head (FCN score layers, dec/disp1) named exactly makes its INPUT tensor one fp16 plane (the producer writes, the head reads half the
 * bytes); heads otherwise follow the format the convolutions give their input. */
sd_status sd_create_with_plan(sd_handle** out, int device, int H, int W, int max_batch, sd_encoder enc, const char* fcn_f16_layers,
                              const char* mono_f16_layers);
const char* sd_default_plan(sd_net net);      /* (monodepth: the ResNet-50 plan; the vgg encoder's default plan is empty) */
/* layers_out (nullable): the 2-product layers of the handle's plan, comma-separated; flop_share_out (nullable): their share of
 * the network's algorithmic FLOPs */
sd_status sd_precision_plan(const sd_handle* h, sd_net net, char* layers_out, size_t cap, double* flop_share_out);
sd_status sd_destroy(sd_handle* h);
const char* sd_last_error(const sd_handle* h);

/* ---------------------------------------------------------------- memory + weights
 * replaces SegmentFrame.restore_model / DepthFrame.restore_model (semantic_depth.py:498-541, :627-653) */
sd_status sd_query_memory(const sd_handle* h, size_t* fcn_weight_bytes, size_t* mono_weight_bytes, size_t* workspace_bytes);
sd_status sd_bind_memory(sd_handle* h, void* fcn_weights_dev, void* mono_weights_dev, void* workspace_dev);
int sd_weight_count(const sd_handle* h, sd_net net);
/* name_out: >= 64 bytes; shape_out: 4 x int64 in TensorFlow layout (conv HWIO, transposed conv HWOI, bias [C]) */
sd_status sd_weight_info(const sd_handle* h, sd_net net, int index, char* name_out, int64_t* shape_out, int* rank_out);
/* data_host: float32, TensorFlow layout; re-laid-out for the kernels and copied into the bound arena (synchronous) */
sd_status sd_load_weight(sd_handle* h, sd_net net, const char* name, const float* data_host, const int64_t* shape, int rank);

/* ---------------------------------------------------------------- the three operators + fused tail */
/* SegmentFrame.segment_frame, semantic_depth.py:544-571 (seq:459-485), for B frames at once.
 * frames: u8 [B,H,W,3].  Outputs (each nullable): logits f32 [B,H,W,3] ('logits:0', fcn8s/fcn.py:241),
 * road/fence u8 [B,H,W] = softmax > 0.5 (:555-556,:563-564), argmax u8 [B,H,W] (fcn8s/fcn.py:218-224). */
sd_status sd_fcn8s_forward(sd_handle* h, const uint8_t* frames, int B, float* logits, uint8_t* road_mask,
                           uint8_t* fence_mask, uint8_t* argmax, void* stream);

/* DepthFrame.compute_disparity, semantic_depth.py:667-678 (seq:568-579), for B frames: /255, (frame, fliplr(frame))
 * pair through monodepth, disp_left_est[0], post_processing (:656-664).  disp_pp: f32 [B,H,W] in fraction of
 * image width.  disp_raw (nullable): f32 [B,2,H,W] = the net's channel-0 output for frame and flipped frame.
 * disp_pp may be NULL when the post-processing is left to sd_postprocess_fuse_backproject (needs B <= the handle's pass size of
 * 32 frames, or disp_raw). */
sd_status sd_monodepth_forward(sd_handle* h, const uint8_t* frames, int B, float* disp_pp, float* disp_raw, void* stream);

/* Input stage, semantic_depth.py:111 / seq:128: cv2.resize(frame, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC) for B
 * uint8 HWC frames already in device memory (OpenCV's scalar fixed-point path: A = -0.75, weights cvRound(w*2048), borders
 * replicated, (sum + 2^21) >> 22).  dst_h, dst_w at most 16384 (down- or up-scaling: semantic_depth.py:341 resizes the overlay back to
 * the original frame size the same way); equal sizes copy. */
sd_status sd_resize_cubic_u8(sd_handle* h, const uint8_t* src, int B, int src_h, int src_w, int channels, uint8_t* dst, int dst_h,
                             int dst_w, void* stream);

/* HOST helper of the frame reader that replaces cv2.imread (semantic_depth.py:105; seq:123): reconstructs the scanlines of an
 * inflated 8-bit non-interlaced PNG (filter byte + width*channels bytes per row; channels 1 gray, 2 gray+alpha, 3 RGB, 4 RGBA)
 * and writes OpenCV's IMREAD_COLOR layout, u8 [height,width,3] BGR (alpha dropped, gray replicated).  Both pointers are HOST
 * memory; no handle, no GPU (semantic_depth_amd/frame_io.py inflates with zlib and calls this from a thread pool). */
sd_status sd_png_unfilter_bgr(const uint8_t* filtered_host, int height, int width, int channels, uint8_t* bgr_out_host);

/* HOST: a whole PNG file (8-bit, non-interlaced; gray, gray+alpha, RGB, RGBA, palette) -> cv2.imread(path) = u8 [height,width,3] BGR:
 * chunk walk, zlib inflate, scanline reconstruction, channel shuffle, palette expansion in one native call (no interpreter lock held).
 * bgr_out_host NULL: only *height_out / *width_out are written (size query).  SD_ERR_INVALID: not such a PNG / corrupt / buffer too small. */
sd_status sd_png_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out, int* width_out);
/* HOST: a baseline / extended-sequential / progressive Huffman JPEG (SOF0 / SOF1 / SOF2, 8-bit; gray or YCbCr with 4:4:4, 4:2:2 or 4:2:0 chroma; restart intervals) ->
 * cv2.imread(path): libjpeg's default decode path restated (jidctint "ISLOW" inverse DCT, "fancy" triangle chroma upsampling, jdcolor's
 * fixed-point YCbCr -> RGB) followed by the EXIF orientation OpenCV's imread applies; u8 [height,width,3] BGR.  The reference's own example
 * frames are JPEGs (assets/images/test_munich/test_3.jpg, semantic_depth.py:105).  Same calling convention as sd_png_decode_bgr;
 * *height_out / *width_out are the dimensions AFTER the orientation.  Arithmetic-coded / lossless / 12-bit / CMYK files, Huffman tables that
 * are not a prefix code and files with more than one frame header: SD_ERR_INVALID. */
sd_status sd_jpeg_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out, int* width_out);
/* HOST: either of the two, by file signature */
sd_status sd_image_decode_bgr(const uint8_t* file_host, size_t len, uint8_t* bgr_out_host, size_t out_capacity, int* height_out, int* width_out);
/* HOST: the batch reader behind frame_io.FrameFeeder (the loop over sorted(glob(...)) of seq:689-701, `cv2.imread` at seq:123): reads and
 * decodes n PNG / JPEG files of height x width on `threads` native threads (<= 0: one per host CPU) into out_host + i * frame_stride (e.g. a
 * pinned staging buffer).  status_out (nullable, int[n]): per-file sd_status (SD_ERR_NOTFOUND: unreadable file; SD_ERR_INVALID: not a
 * PNG of that shape).  Returns SD_OK when every file decoded. */
sd_status sd_decode_files_bgr(const char* const* paths, int n, int height, int width, uint8_t* out_host, size_t frame_stride, int threads,
                              int* status_out);

/* HOST helper of the PLY writer that replaces semantic_depth_lib/point_cloud_2_ply.py:70 (numpy.savetxt(fh, rows, "%f %f %f %d %d %d")):
 * n vertex rows "x y z r g b\n" -- coordinates as "%f" % float(v) prints them (fixed, six decimals, correctly rounded; nan / inf /
 * -inf), colours as integers -- into out[0 .. cap).  xyz f64 [n,3], rgb int64 [n,3], HOST memory; threads <= 0: one per core, at most
 * 16.  Returns the number of bytes written, or SD_ERR_INVALID (bad argument, or cap too small: cap / n bytes must hold any row). */
int64_t sd_ply_format_rows(const double* xyz_host, const int64_t* rgb_host, int64_t n, char* out_host, int64_t cap, int threads);

/* DepthFrame.post_processing alone, semantic_depth.py:656-664: disp_raw f32 [B,2,H,W] -> disp_pp f32 [B,H,W] */
sd_status sd_post_process(sd_handle* h, const float* disp_raw, int B, float* disp_pp, void* stream);

/* disparity scaling + DepthFrame.compute_3D_points + BGR->RGB + mask gather,
 * semantic_depth.py:145,160-161,183-187,686-697 (seq:146,152-153,170-174), for B frames.
 * disp_pp f32 [B,H,W]; masks u8 [B,H,W]; frames u8 [B,H,W,3]; cams[B] (HOST pointer).
 * points_dense (nullable) f32 [B,H,W,3] = cv2.reprojectImageTo3D(disp_pp*mult, Q).
 * road_xyz f32 [B,cap,3], road_rgb u8 [B,cap,3] (nullable), n_road i32 [B]; same for fence (all nullable as a group).
 * cap = per-frame capacity in points (H*W is always enough); order = row-major order of the True pixels. */
sd_status sd_fuse_backproject(sd_handle* h, const float* disp_pp, const uint8_t* road_mask, const uint8_t* fence_mask,
                              const uint8_t* frames, const sd_camera* cams_host, int B, int cap, float* points_dense,
                              float* road_xyz, uint8_t* road_rgb, int32_t* n_road, float* fence_xyz, uint8_t* fence_rgb,
                              int32_t* n_fence, void* stream);

/* DepthFrame.post_processing (:656-664) + the above in ONE pass over the pixels: the raw pair is read once, disp_pp_out
 * (f32 [B,H,W], required) is written once and never re-read, both clouds are gathered in the same launch (decoupled look-back
 * compaction).  disp_raw f32 [B,2,H,W], or NULL = the raw output of the handle's last sd_monodepth_forward (B <= 32 frames).
 * Same results as sd_post_process followed by sd_fuse_backproject, bit for bit. */
sd_status sd_postprocess_fuse_backproject(sd_handle* h, const float* disp_raw, float* disp_pp_out, const uint8_t* road_mask,
                                          const uint8_t* fence_mask, const uint8_t* frames, const sd_camera* cams_host, int B, int cap,
                                          float* road_xyz, uint8_t* road_rgb, int32_t* n_road, float* fence_xyz, uint8_t* fence_rgb,
                                          int32_t* n_fence, void* stream);

/* the road chain of FrameProcessor.process_frame, semantic_depth.py:203-259 (seq:180-238):
 * z-cut -> MAD(y) -> MAD(x) -> plane fit -> [Open3D statistical + radius] -> end points -> width.
 * road_xyz f32 [B,cap,3], n_road i32 [B] (device).  road_rgb (nullable) u8 [B,cap,3]: the colours the reference carries
 * through every filter (road_colors, :206-245).  results: DEVICE array of B sd_rw_result.
 * final_xyz (nullable) f32 [B,cap,3] receives the denoised cloud, final_rgb (nullable, needs road_rgb) its colours,
 * n_final i32 [B] its size. */
sd_status sd_road_width(sd_handle* h, const float* road_xyz, const uint8_t* road_rgb, const int32_t* n_road, int B, int cap,
                        const sd_rw_params* params_host, sd_rw_result* results, float* final_xyz, uint8_t* final_rgb,
                        int32_t* n_final, void* stream);

/* fence chain + fence-to-fence distance, semantic_depth.py:273-334 (seq:245-298), for B frames:
 * MAD(y, mad_y) -> |z| < z_max -> extract_pcls at mean x -> left: MAD(x, mad_left) + plane(axis 0, plane_thr);
 * right: MAD(x, mad_right) + plane(axis 0, plane_thr) -> both planes intersected with the road plane at z = -depth
 * -> Euclidean distance.  road: DEVICE array of B sd_rw_result (their .plane is the road plane, from sd_road_width). */
typedef struct {
    double depth;        /* :325 z=self.depth (10.0) */
    double mad_y;        /* remove_noise_by_mad(fence, 1, 5.0)      :279-280 */
    double z_max;        /* threshold_complete(fence, 2, 35.0)      :283-284 */
    double mad_left;     /* remove_noise_by_mad(left, 0, 5.0)       :291 */
    double mad_right;    /* remove_noise_by_mad(right, 0, 1.0)      :302 */
    double plane_thr;    /* remove_noise_by_fitting_plane(axis=0, threshold=1.0) :294-298, :305-309 */
} sd_f2f_params;
typedef struct {
    double dist;                       /* dist_f2f, :327 */
    double left_pt[3], right_pt[3];    /* plane intersections at z = -depth, :321-326 */
    double plane_left[4], plane_right[4];
    int32_t counts[7];                 /* n_fence, after MAD(y), after |z| cut, left, right, left final, right final */
    int32_t ok;                        /* 0 when a side is empty / the planes are degenerate (dist is NaN) */
} sd_f2f_result;
/* fence_rgb (nullable) u8 [B,cap,3]; left_* / right_* (nullable) receive the denoised left / right fence clouds the
 * reference writes to <name>_FENCE.ply (:412-415): xyz f32 [B,cap,3], rgb u8 [B,cap,3], sizes = results[b].counts[5], [6] */
sd_status sd_fence_to_fence(sd_handle* h, const float* fence_xyz, const uint8_t* fence_rgb, const int32_t* n_fence, int B, int cap,
                            const sd_rw_result* road, const sd_f2f_params* params_host, sd_f2f_result* results,
                            float* left_xyz, uint8_t* left_rgb, float* right_xyz, uint8_t* right_rgb, void* stream);

/* ---------------------------------------------------------------- pcl.py, function by function
 * (semantic_depth_lib/pcl.py; one cloud per call: xyz f32 [n,3], rgb u8 [n,3] nullable, n on the host).
 * Outputs keep the input row order.  *n_out is a DEVICE int32. */
/* pcl.remove_from_to, pcl.py:30-43 (keeps coord[axis] < -to_meter) */
sd_status sd_pcl_remove_from_to(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double to_meter,
                                float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream);
/* pcl.remove_noise_by_mad + mad, pcl.py:46-81.  stats_out (nullable, device f32[2]) = {median, MAD} */
sd_status sd_pcl_remove_noise_by_mad(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double threshold,
                                     float* xyz_out, uint8_t* rgb_out, int32_t* n_out, float* stats_out, void* stream);
/* pcl.remove_noise_by_fitting_plane, pcl.py:84-209 (without the visualisation grid).  coeff_out: device f64[4] = Cx,Cy,Cz,C */
sd_status sd_pcl_remove_noise_by_fitting_plane(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis,
                                               double threshold, float* xyz_out, uint8_t* rgb_out, int32_t* n_out,
                                               double* coeff_out, void* stream);
/* pcl.threshold_complete, pcl.py:240-250 (keeps |coord[axis]| < threshold) */
sd_status sd_pcl_threshold_complete(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, double threshold,
                                    float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream);
/* pcl.extract_pcls, pcl.py:253-268: split at np.mean(coord[axis]) (reproduced bit for bit): left = coord < mean,
 * right = coord > mean.  mean_out (nullable): device f32[1] */
sd_status sd_pcl_extract_pcls(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int axis, float* left_xyz,
                              uint8_t* left_rgb, int32_t* n_left, float* right_xyz, uint8_t* right_rgb, int32_t* n_right,
                              float* mean_out, void* stream);
/* pcl.get_end_points_of_road, pcl.py:271-313: first min-x and max-x rows of the depth window.
 * out: device sd_rw_result (only found, x_left, x_right, left_pt, right_pt, width are written) */
sd_status sd_pcl_get_end_points_of_road(sd_handle* h, const float* xyz, int n, double depth, double window,
                                        sd_rw_result* out, void* stream);
/* Open3D statistical_outlier_removal / radius_outlier_removal as called at semantic_depth.py:234-241 */
sd_status sd_o3d_statistical_outlier_removal(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int nb_neighbors,
                                             double std_ratio, float* xyz_out, uint8_t* rgb_out, int32_t* n_out,
                                             double* mean_dist_out /* nullable, device f64[n] */, void* stream);
sd_status sd_o3d_radius_outlier_removal(sd_handle* h, const float* xyz, const uint8_t* rgb, int n, int nb_points,
                                        double radius, float* xyz_out, uint8_t* rgb_out, int32_t* n_out, void* stream);

/* ---------------------------------------------------------------- introspection (tests / profiling) */
/* copy an intermediate activation of the last forward (e.g. "layer3_out", "conv5") to out (device f32);
 * numel_out receives its element count; shape_out 4 x int64 [N,H,W,C] */
sd_status sd_net_tensor(sd_handle* h, sd_net net, const char* name, float* out, size_t out_capacity_floats,
                        int64_t* shape_out, void* stream);
/* per-kernel timing of the conv engine with HIP events on the launch stream (bench.py roofline).
 * sd_profile(h,1) starts recording an event pair around every conv launch; sd_profile_read synchronises the
 * device, sums elapsed time / algorithmic FLOPs (2*M*N*K) / launches per kernel instantiation into out[0..*n),
 * and clears the recording.  out: HOST array of capacity cap_buckets. */
typedef struct {
    char kernel[64];
    int64_t launches;
    double ms;
    double flops;
    double bytes;      /* algorithmic HBM bytes of those launches: every source tensor, the weights and the output tensor ONCE, in the
                          formats the engine stores them in (what a launch cannot avoid moving; the PMC traffic is compared with it) */
} sd_profile_bucket;
sd_status sd_profile(sd_handle* h, int enable);
sd_status sd_profile_read(sd_handle* h, sd_profile_bucket* out_host, int cap_buckets, int* n_out);

/* number of conv-engine FLOPs (2*M*N*K over all layers, per image) of a plan — for roofline accounting */
double sd_net_flops_per_image(const sd_handle* h, sd_net net);

/* frames of one network pass (the chunk size latched at sd_create: min(max_batch, SEMDEPTH_CHUNK or 32)); a batch of at most this many
 * frames leaves its raw disparity pair in the arena for sd_postprocess_fuse_backproject(disp_raw = NULL) */
int sd_pass_frames(const sd_handle* h);

/* fp16 range guard of the reduced-precision plans: the conv epilogues that write fp16 planes clamp at +-65504 and COUNT the values
 * they clamped (and NaNs) in a device counter.  *count_out = values clamped since the handle was bound or the counter last reset
 * (synchronises the device); reset != 0 clears it.  A non-zero count means the plan does not fit these weights / inputs: run the
 * layer's producer with more products, or the exact engine (the reference has no such failure mode: it computes in f32). */
sd_status sd_saturation_count(sd_handle* h, uint64_t* count_out, int reset);
/* Tail overlap: the conv kernels of the 3x3 layers are persistent workgroups that fill every CU's register file, so work on a second stream
 * (the per-frame tail of the previous batch: back-projection, road chain) finds no CU while they run.  n > 0 makes those launches use
 * (CUs - n) workgroups; the n CUs left free take the side stream's kernels.  0 (default, or SEMDEPTH_RESERVE_CUS at sd_create) = every CU.
 * Results do not depend on it (the tiles a workgroup walks change, not their arithmetic). */
sd_status sd_set_reserved_cus(sd_handle* h, int n);

/* the same count without a device synchronisation: an 8-byte device-to-host copy enqueued on `stream` behind the work already on it
 * (host_dst = pinned host memory of the caller).  The host-side classes use it to turn a range violation into an ERROR of the call that
 * produced it instead of a counter somebody has to poll (semantic_depth_amd/engine.py Engine.check_range). */
sd_status sd_saturation_count_async(sd_handle* h, uint64_t* host_dst, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMDEPTH_H */
