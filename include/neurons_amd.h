/*
 * neurons_amd.h — C ABI of libneurons_amd.so (MI355X / gfx950 only).
 *
 * The reference (xmed-lab/NEURONS) is pure Python: it has no FFI, so this header is the boundary a
 * maintainer would bind from Python (ctypes stubs in INTEGRATION.md; the shipped binding is
 * neurons_amd/_lib.py).  Each entry point names the reference interface it replaces.
 *
 * Conventions
 *   - Every function returns nr_status (0 = NR_OK).  On failure nr_last_error() returns a message;
 *     nothing is thrown across the boundary.
 *   - All pointers named *_dev are device pointers owned by the caller (PyTorch allocations).  The
 *     library owns: converted weights, one workspace arena sized at nr_net_plan(), a captured hipGraph.
 *   - All work is enqueued on the hipStream_t passed in; no synchronisation except in nr_net_plan()
 *     and nr_net_load_tensor().  A handle is not thread-safe; one handle per GPU / process.
 *   - Activation tensors crossing this ABI between the two networks (ControlNet residuals) are
 *     channels-last bf16:  [2B*F][h][w][C], frame-image index n = b*F + f.
 *   - Latent-space tensors (sample, eps, controlnet_cond, mask) are fp32 NCFHW, exactly the
 *     reference's torch layout "b c f h w".
 */
#ifndef NEURONS_AMD_H
#define NEURONS_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int nr_status;
#define NR_OK 0
#define NR_ERR_ARG 1
#define NR_ERR_STATE 2
#define NR_ERR_MISSING_WEIGHT 3
#define NR_ERR_HIP 4
#define NR_ERR_UNSUPPORTED 5

typedef void* nr_stream; /* hipStream_t */
typedef struct nr_net nr_net;

#define NR_KIND_UNET3D 0     /* animatediff/models/unet.py:38 UNet3DConditionModel            */
#define NR_KIND_SPARSECTRL 1 /* animatediff/models/sparse_controlnet.py:85 SparseControlNetModel */
#define NR_KIND_SGM_UNET 2   /* generative_models/sgm/modules/diffusionmodules/openaimodel.py:472 UNetModel (unCLIP keyframes) */

#define NR_KIND_VAE_DECODER 3 /* generative_models/sgm/modules/diffusionmodules/model.py:612 Decoder behind
                                 sgm/models/autoencoder.py:490 decode (= diffusers AutoencoderKL.decode)     */
#define NR_KIND_VAE_ENCODER 4 /* generative_models/sgm/modules/diffusionmodules/model.py:495 Encoder + quant_conv
                                 (sgm/models/autoencoder.py:468-488 = diffusers AutoencoderKL.encode)         */
#define NR_KIND_CLIP_TEXT 5   /* transformers CLIPTextModel as called by _encode_prompt (pipeline_neuroclips.py:153-240) */

/* test hooks: ONE reference module as a network of its own (see nr_leaf_forward below) */
#define NR_KIND_LEAF_TRANSFORMER3D 6 /* animatediff/models/attention.py:31 Transformer3DModel (one BasicTransformerBlock)          */
#define NR_KIND_LEAF_TEMPORAL 7      /* animatediff/models/motion_module.py:48 VanillaTemporalModule -> TemporalTransformer3DModel */

#define NR_MAX_LEVELS 4

/* Mirrors the constructor arguments that reach the hot path (unet.py:42-90; SD-1.5 unet/config.json +
 * configs/inference/inference-v3.yaml; sparse_controlnet.py:88-140 +
 * configs/inference/sparsectrl/latent_condition.yaml). */
typedef struct nr_net_config {
  int32_t kind;                              /* NR_KIND_*                                         */
  int32_t in_channels;                       /* 4                                                 */
  int32_t out_channels;                      /* 4 (UNet only)                                     */
  int32_t num_levels;                        /* len(block_out_channels) = 4                       */
  int32_t block_out_channels[NR_MAX_LEVELS]; /* 320,640,1280,1280                                 */
  int32_t down_block_has_attn[NR_MAX_LEVELS]; /* CrossAttnDownBlock3D=1, DownBlock3D=0  (1,1,1,0) */
  int32_t up_block_has_attn[NR_MAX_LEVELS];  /* UpBlock3D=0, CrossAttnUpBlock3D=1       (0,1,1,1) */
  int32_t layers_per_block;                  /* 2                                                 */
  int32_t num_heads;                         /* "attention_head_dim" = 8 = number of heads        */
  int32_t cross_attention_dim;               /* 768                                               */
  int32_t norm_num_groups;                   /* 32                                                */
  float norm_eps;                            /* 1e-5                                              */
  int32_t use_motion_module;                 /* 1                                                 */
  int32_t motion_num_heads;                  /* 8                                                 */
  int32_t motion_num_attention_blocks;       /* 2 (UNet v3) / 1 (SparseCtrl)                      */
  int32_t motion_pe_max_len;                 /* 24 (UNet default) / 32 (SparseCtrl)               */
  int32_t motion_module_mid_block;           /* 0                                                 */
  int32_t conditioning_channels;             /* SparseCtrl: 4 (+1 mask channel is implied)        */
  int32_t set_noisy_sample_input_to_zero;    /* SparseCtrl: 1                                     */
  /* sgm UNetModel only (generative_models/configs/unclip6.yaml:47-63); for that kind block_out_channels[i] =
   * channel_mult[i]*model_channels, layers_per_block = num_res_blocks, down_block_has_attn[i] = (2^i in
   * attention_resolutions), cross_attention_dim = context_dim                                                  */
  int32_t transformer_depth[NR_MAX_LEVELS];  /* 1,2,10 (0 for the other kinds = depth 1)          */
  int32_t num_head_channels;                 /* 64: heads = C / 64 (0 for the other kinds)        */
  int32_t adm_in_channels;                   /* 1024: width of the `y` vector                     */
} nr_net_config;

#define NR_DTYPE_F32 0
#define NR_DTYPE_BF16 1

/* ---- lifecycle ------------------------------------------------------------------------------ */

/* replaces UNet3DConditionModel.__init__ / SparseControlNetModel.__init__ */
nr_status nr_net_create(const nr_net_config* cfg, nr_net** out);
void nr_net_destroy(nr_net* h);
const char* nr_last_error(void);

/* replaces nn.Module.load_state_dict (animatediff/utils/util.py:120,139-144): one call per
 * state-dict entry, reference key names, host fp32 data ([ndim] shape).  Unknown keys are kept and
 * ignored; missing keys are reported at nr_net_plan(). */
nr_status nr_net_load_tensor(nr_net* h, const char* key, const float* host_data, const int64_t* shape, int32_t ndim);

/* Fix the problem size: batch = CFG-expanded batch (2B; <= 64: the grouped SparseCtrl schedule plans G x 2B), frames F, latent h x w, ctx_len =
 * tokens of encoder_hidden_states (77 for CLIP).  Converts/uploads
 * weights (first call), builds the launch plan and allocates the workspace arena.  May be called
 * again with another shape. */
nr_status nr_net_plan(nr_net* h, int32_t batch, int32_t frames, int32_t lat_h, int32_t lat_w, int32_t ctx_len);

/* Frees the host fp32 copies of the state dict (7 GB for the two full-size networks); the converted
 * device weights stay.  Re-planning with another shape keeps working; loading new tensors re-converts. */
nr_status nr_net_release_host_weights(nr_net* h);

/* The cross-attention context (encoder_hidden_states / "crossattn") is constant over the denoising steps of a clip:
 * its bf16 copy and every to_k|to_v projection are computed on the first forward and reused until this is called.
 * Call it whenever the CONTENTS behind ctx_dev change (a new pointer value alone is not detected). */
nr_status nr_net_invalidate_context(nr_net* h);

/* 1: run forward as a captured hipGraph (re-captured when an I/O pointer changes); 0: eager launches */
nr_status nr_net_set_graph(nr_net* h, int32_t enable);

/* bytes of workspace / converted weights held by the handle */
int64_t nr_net_workspace_bytes(const nr_net* h);
int64_t nr_net_weight_bytes(const nr_net* h);
int32_t nr_net_num_residuals(const nr_net* h);                     /* 12 (+1 mid) for the default config */
nr_status nr_net_residual_shape(const nr_net* h, int32_t i, int32_t* C, int32_t* hh, int32_t* ww); /* i == num -> mid */

/* ---- forward -------------------------------------------------------------------------------- */

/* replaces UNet3DConditionModel.forward (unet.py:320-475)
 *   sample_dev   fp32 [batch][4][F][h][w]
 *   timesteps    host fp32 [batch]  (the reference broadcasts a scalar: unet.py:371-384)
 *   ctx_dev      fp32 [batch][ctx_len][cross_attention_dim]  encoder_hidden_states
 *   down_res_dev NULL or num_residuals pointers, bf16 channels-last (down_block_additional_residuals)
 *   mid_res_dev  NULL or bf16 channels-last                       (mid_block_additional_residual)
 *   out_dev      fp32 [batch][4][F][h][w]                          (.sample)                        */
nr_status nr_unet3d_forward(nr_net* h, nr_stream stream, const float* sample_dev, const float* timesteps,
                            const float* ctx_dev, int32_t ctx_len, const void* const* down_res_dev,
                            const void* mid_res_dev, float* out_dev);

/* replaces SparseControlNetModel.forward (sparse_controlnet.py:450-581)
 *   sample_dev   fp32 [batch][4][F][h][w]; may be NULL when set_noisy_sample_input_to_zero
 *   cond_dev     fp32 [cond_batch][cond_ch][F][h][w]  controlnet_cond
 *   mask_dev     fp32 [cond_batch][1][F][h][w]        conditioning_mask
 *   cond_batch   batch of cond/mask; broadcast over the CFG halves as b % cond_batch
 *   scale        conditioning_scale
 *   out_down_dev num_residuals pointers, bf16 channels-last, written
 *   out_mid_dev  bf16 channels-last, written                                                        */
nr_status nr_sparsectrl_forward(nr_net* h, nr_stream stream, const float* sample_dev, const float* timesteps,
                                const float* ctx_dev, int32_t ctx_len, const float* cond_dev, const float* mask_dev,
                                int32_t cond_batch, float scale, void* const* out_down_dev, void* out_mid_dev);

/* One denoising-step network evaluation = SparseControlNetModel.forward followed by UNet3DConditionModel.forward
 * (pipeline_neuroclips.py:460-475) in ONE call: SparseCtrl runs on its own stream concurrently with the U-Net's
 * encoder + mid block (which do not depend on it); the residual adds and the decoder wait for it.  Same results as
 * nr_sparsectrl_forward + nr_unet3d_forward.  res_*_dev: caller-owned channels-last bf16 residual buffers
 * (written by SparseCtrl, read by the U-Net).  Requires set_noisy_sample_input_to_zero (sample is not read by SparseCtrl).
 * next_timesteps (host fp32 [batch], may be NULL): the timesteps of the FOLLOWING denoising step of the same clip.
 * With the noisy sample zeroed, SparseCtrl depends on (timestep, context, condition) only, so its next evaluation is
 * issued as soon as this step's residual adds have consumed the current one and overlaps this step's decoder; the
 * next call uses it if its timesteps / context / cond / mask / residual pointers are identical (else it re-runs).
 * Contract when non-NULL: ctx/cond/mask/res buffers stay alive and unmodified until the next call on these handles
 * or nr_net_invalidate_context(ctrl).                                                                             */
nr_status nr_denoise_step_forward(nr_net* unet, nr_net* ctrl, nr_stream stream, const float* sample_dev,
                                  const float* timesteps, const float* ctx_dev, int32_t ctx_len, const float* cond_dev,
                                  const float* mask_dev, int32_t cond_batch, float scale, void* const* res_down_dev,
                                  void* res_mid_dev, float* out_dev, const float* next_timesteps);

/* replaces OpenAIWrapper.forward -> UNetModel.forward (sgm/modules/diffusionmodules/wrappers.py:23-34,
 * openaimodel.py:816-853).  x_dev fp32 [batch][4][h][w] is multiplied by in_scale (= c_in of
 * denoiser.py:36-39) on the fly; timesteps = c_noise (host fp32 [batch]); ctx_dev fp32 [batch][ctx_len][context_dim]
 * ("crossattn"); y_dev fp32 [batch][adm_in_channels] ("vector"); out_dev fp32 [batch][4][h][w].               */
nr_status nr_sgm_unet_forward(nr_net* h, nr_stream stream, const float* x_dev, float in_scale, const float* timesteps,
                              const float* ctx_dev, int32_t ctx_len, const float* y_dev, float* out_dev);

/* replaces DiffusionEngine.decode_first_stage (sgm/models/diffusion.py:118-135: z / scale_factor ->
 * AutoencodingEngineLegacy.decode, autoencoder.py:490-494 = post_quant_conv -> Decoder.forward, model.py:723-757) and
 * the per-frame vae.decode of AnimationPipeline.decode_latents (pipeline_animation.py:243-256; same network under
 * diffusers parameter names, see neurons_amd/vae.py for the key map).  Config: kind NR_KIND_VAE_DECODER,
 * in_channels = z_channels (= embed_dim), out_channels = 3, block_out_channels = ch * ch_mult, layers_per_block =
 * num_res_blocks, norm_num_groups 32, norm_eps 1e-6; plan with (batch = images, frames = 1, h, w, ctx_len = 0).
 *   z_dev fp32 [batch][z][h][w]; z_scale = 1 / scale_factor; out_dev fp32 [batch][3][8h][8w] =
 *   decoder(z) * out_mul + out_add, clamped to [0, 1] when clamp01 != 0: the callers' image post-scaling fused into
 *   the last kernel ((x / 2 + 0.5).clamp(0, 1) pipeline_animation.py:252; clamp(x * .8 + .2, 0, 1) utils.py:348);
 *   (1, 0, 0) returns the raw decoder output.                                                                     */
nr_status nr_vae_decode(nr_net* h, nr_stream stream, const float* z_dev, float z_scale, float out_mul, float out_add,
                        int32_t clamp01, float* out_dev);

/* replaces text_encoder(text_input_ids, attention_mask=None)[0] in _encode_prompt (pipeline_neuroclips.py:197-200,
 * 231-234): transformers CLIPTextModel -> last_hidden_state (CLIPTextEmbeddings, 12 pre-LN encoder layers with causal
 * self-attention and quick_gelu MLP, final_layer_norm).  Config: kind NR_KIND_CLIP_TEXT, block_out_channels[0] =
 * hidden_size (768), num_heads (12), layers_per_block = num_hidden_layers (12), cross_attention_dim = intermediate_size
 * (3072), in_channels = vocab_size (49408), motion_pe_max_len = max_position_embeddings (77), norm eps 1e-5;
 * plan with (batch, 1, 1, seq_len, 0).  State-dict keys: transformers' (text_model.embeddings.token_embedding.weight ...).
 *   ids_dev int32 [batch][seq_len] (device); out_dev fp32 [batch][seq_len][hidden].  Tokenisation stays with the caller. */
nr_status nr_clip_text_forward(nr_net* h, nr_stream stream, const int32_t* ids_dev, float* out_dev);

/* replaces vae.encode(2 * x - 1).latent_dist (scripts/neuroclips_video.py:267,282, scripts/neuroclips_video_enhance.py:
 * 268,283) = AutoencodingEngine.encode up to the moments (sgm/models/autoencoder.py:468-488: Encoder.forward
 * model.py:584-609 -> quant_conv).  Config: kind NR_KIND_VAE_ENCODER, in_channels 3, out_channels = 2 * z_channels = 8,
 * block_out_channels = ch * ch_mult, layers_per_block = num_res_blocks, norm eps 1e-6; plan with (batch = images,
 * frames = 1, IMAGE h, IMAGE w, ctx_len = 0).
 *   x_dev fp32 [batch][3][h][w]; the network sees x * in_mul + in_add (the callers' 2 * x - 1);
 *   moments_dev fp32 [batch][8][h/8][w/8] = (mean | logvar), the DiagonalGaussianDistribution parameters.          */
nr_status nr_vae_encode(nr_net* h, nr_stream stream, const float* x_dev, float in_mul, float in_add, float* moments_dev);

/* DiagonalGaussianDistribution.sample() / .mode() (sgm/modules/distributions/distributions.py:24-42,71-72) times the
 * callers' latent scale: out[n][z][hw] = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale.
 * noise_dev fp32 [n][z][hw] is drawn by the caller (RNG order is the caller's contract); null = mode().          */
nr_status nr_gaussian_sample(nr_stream stream, const float* moments_dev, const float* noise_dev, float* out_dev, int32_t n,
                             int32_t z_channels, int32_t hw, float scale);

/* replaces Denoiser.forward's output scaling + VanillaCFG + EulerEDMSampler.sampler_step with s_churn = 0
 * (denoiser.py:36-39, denoiser_scaling.py:29-37, guiders.py:28-31, sampling_utils.py:34-35, sampling.py:98-112):
 *   den_k = net_k * (-sigma_quantized) + x ;  den = den_u + scale * (den_c - den_u) ;  d = (x - den) / sigma ;
 *   x_out = x + d * (sigma_next - sigma).     net_dev fp32 [2B][...] (uncond first), x fp32 [B][...], n = numel(x);
 *   sigma_quantized = nearest table sigma the DiscreteDenoiser evaluates the network at (denoiser.py:61-75)    */
nr_status nr_edm_cfg_euler_step(nr_stream stream, const float* net_dev, const float* x_dev, float* x_out_dev, int64_t n,
                                float cfg_scale, float sigma_quantized, float sigma, float sigma_next);

/* replaces one iteration of BrainDiffusionPrior.p_sample_loop_ddpm (model_variants/BrainModel_neurons.py:324-341,363-389) behind the network
 * call, i.e. DiffusionPrior.p_mean_variance + the ancestral update of dalle2-pytorch 1.15.6 (requirements.txt:11; NOT vendored: parity
 * unpinned): [pred = null + (pred - null) * cond_scale]; x_start = pred (mode 0, predict_x_start: the NEURONS prior) | sqrt(ac) x -
 * sqrt(1 - ac) pred (mode 1, v) | sqrt(1 / ac) x - sqrt(1 / ac - 1) pred (mode 2, noise; clamp = clip_denoised); x_out = coef1 x_start +
 * coef2 x + exp(0.5 * posterior_log_variance_clipped) * noise, noise_dev == NULL at t = 0.  The per-t coefficients are formed here from
 * (alphas_cumprod[t], alphas_cumprod[t - 1] (1 at t = 0), betas[t]) exactly as NoiseScheduler.__init__ does.  All tensors fp32 [n];
 * x_start_out_dev may be NULL; x_out_dev may alias x_dev. */
nr_status nr_prior_p_sample_step(nr_stream stream, const float* pred_dev, const float* pred_null_dev, const float* x_dev, const float* noise_dev,
                                 float* x_out_dev, float* x_start_out_dev, int64_t n, float cond_scale, int32_t mode, int32_t clamp,
                                 double alpha_cumprod_t, double alpha_cumprod_prev, double beta_t);

/* replaces the CFG combine + DDIMScheduler.step (pipeline_neuroclips.py:478-483; diffusers 0.11.1 DDIM eta=0)
 *   eps_dev fp32 [2B or B][...], x_dev fp32 [B][...] -> x_out_dev (may alias x_dev); n = elements of x     */
nr_status nr_cfg_ddim_step(nr_stream stream, const float* eps_dev, const float* x_dev, float* x_out_dev, int64_t n,
                           float guidance_scale, int32_t do_cfg, double alpha_prod_t, double alpha_prod_t_prev);

/* replaces the CFG combine alone (pipeline_neuroclips.py:478-480) for a caller that keeps its own scheduler object
 * (scripts/neuroclips_video.py:219) and calls its .step (pipeline_neuroclips.py:483) on the combined noise:
 *   eps_dev fp32 [2B][...] (uncond half first) -> eps_out_dev fp32 [B][...] = e_u + g (e_t - e_u); n = elements of the output */
nr_status nr_cfg_combine(nr_stream stream, const float* eps_dev, float* eps_out_dev, int64_t n, float guidance_scale);

/* ---- measurement ----------------------------------------------------------------------------- */
#define NR_PROF_IGEMM 0     /* MFMA implicit-GEMM conv / Linear kernel            */
#define NR_PROF_GROUPNORM 1
#define NR_PROF_LAYERNORM 2
#define NR_PROF_ATTENTION 3
#define NR_PROF_OTHER 4     /* boundary convs, time embedding, adds, converts     */
#define NR_PROF_KINDS 5
typedef struct nr_profile {
  double ms[NR_PROF_KINDS];       /* summed launch durations, HIP events on the launch stream */
  double flops[NR_PROF_KINDS];    /* algorithmic FLOPs (2*M*N*K; 4*B*H*Lq*Lk*d for attention)   */
  double bytes[NR_PROF_KINDS];    /* algorithmic bytes (inputs + weights + outputs, bf16)       */
  int32_t launches[NR_PROF_KINDS];
} nr_profile;
/* Re-runs the launch plan of the most recent forward eagerly on `stream`, one HIP event pair per
 * launch, and reports per-kernel-class totals (bench.py's roofline object). */
nr_status nr_net_profile_last(nr_net* h, nr_stream stream, nr_profile* out);

/* ---- grouped SparseCtrl schedule (pipeline_neuroclips.py:460-475 evaluated G steps at a time) --------------------------------------
 * With set_noisy_sample_input_to_zero (sparse_controlnet.py:469-470: the NEURONS configuration) SparseCtrl sees the timestep, the text
 * context and the condition, nothing of the denoising state, so its evaluations for G consecutive DDIM steps can run as ONE forward on a
 * handle planned for G x (CFG batch) samples, ahead of the U-Net steps that consume them.
 * nr_sparsectrl_forward_async: like nr_sparsectrl_forward (no `sample`), but on the handle's own stream and NOT joined to `stream`: it
 *   starts after the work already enqueued on `stream` and records its completion in event slot `slot` (0 or 1).  timesteps: one per
 *   planned sample.  The output buffers must stay untouched until a consumer has waited for the slot.
 * nr_unet3d_forward_after: nr_unet3d_forward with residuals, whose residual adds wait for slot `slot` of `ctrl`; the encoder and mid
 *   block run while the SparseCtrl evaluation is still pending.  down_res_dev / mid_res_dev: the slices of the group's outputs that
 *   belong to this step (batch-major buffers: step p of the group starts at sample p x CFG batch). */
nr_status nr_sparsectrl_forward_async(nr_net* ctrl, nr_stream stream, const float* timesteps, const float* ctx_dev, int32_t ctx_len,
                                      const float* cond_dev, const float* mask_dev, int32_t cond_batch, float scale,
                                      void* const* out_down_dev, void* out_mid_dev, int32_t slot);
nr_status nr_unet3d_forward_after(nr_net* unet, nr_net* ctrl, int32_t slot, nr_stream stream, const float* sample_dev,
                                  const float* timesteps, const float* ctx_dev, int32_t ctx_len, const void* const* down_res_dev,
                                  const void* mid_res_dev, float* out_dev);

/* BASELINE config 5: run the spatial self- and text cross-attention cores (motion_module_new.py:258-287) with OCP e4m3 MFMA
 * operands (fp32 softmax statistics and accumulation).  Off by default (bf16); changing it invalidates the plan. */
nr_status nr_net_set_attention_fp8(nr_net* h, int32_t enable);

/* SparseCtrl: tell the engine which frames carry a condition (controlnet_cond / conditioning_mask not all zero there;
 * pipeline_neuroclips.py:447-458 fills only controlnet_image_index).  With set_noisy_sample_input_to_zero every other frame enters the
 * network as the same constant image, so down_blocks[0].resnets[0] + attentions[0] are evaluated on the conditioned frames plus ONE
 * representative and broadcast before the first motion module (sparse_controlnet.py:517-545, unet_blocks.py:382-421): same results, 1/8
 * of the work of those two modules for one condition frame in sixteen.  The CALLER guarantees the zero frames (NativeSparseCtrl derives
 * the list from the tensors themselves).  n < 0 switches back to the full evaluation (default).  Invalidates the plan when it changes. */
nr_status nr_sparsectrl_set_condition_frames(nr_net* h, const int32_t* frames, int32_t n);

/* Batch-independent arithmetic (default off; NR_DETERMINISTIC_BATCH=1 turns it on for new handles): every plan choice that can move a bf16
 * rounding point or an fp32 summation order -- LayerNorm folded into the GEMM vs the separate kernel, split-K depth, row-panel / fused-kernel
 * eligibility, GroupNorm variant and chunking, the per-workgroup weight-stream rotation of the fused kernels -- is made for the rows of ONE
 * clip's CFG pair instead of the rows of the whole call.  A clip then gets the same result alone, in a batch of 8 (BASELINE config 4) or in
 * any SparseCtrl group size; the price is the speed of the shapes that would have taken another plan.  Changing it invalidates the plan. */
nr_status nr_net_set_deterministic_batch(nr_net* h, int32_t enable);
/* How many samples of the batch belong to ONE clip: 2 (default: the classifier-free-guidance pair, pipeline_neuroclips.py:435) or 1 (guidance
 * off, do_classifier_free_guidance false: the batch holds one sample per clip).  Only read in deterministic-batch mode, where "the rows of
 * one clip" is what every plan choice is made for; the pipeline sets it per call.  Changing it in that mode invalidates the plan. */
nr_status nr_net_set_clip_samples(nr_net* h, int32_t samples);
/* Classifier-free-guidance de-duplication (U-Net handles; round 6).  With enable != 0 the caller PROMISES, for every forward until it is cleared,
 * that the second half of the batch repeats the first: sample[b] == sample[b + B/2] and timestep[b] == timestep[b + B/2] -- what the reference's
 * denoising loop feeds (`torch.cat([latents] * 2)` with one timestep, pipeline_neuroclips.py:435); only encoder_hidden_states differ.  The engine
 * then evaluates everything in front of the first cross-attention (conv_in, down_blocks[0].resnets[0], norm / proj_in / norm1 / attn1 of
 * down_blocks[0].attentions[0]: unet.py:395-400, attention.py:256-280) on B/2 samples and broadcasts: exact algebra, the same kernels on half the
 * rows.  Ignored in deterministic-batch mode and with debug taps.  Default off: a caller that passes arbitrary batches (the plain
 * nr_unet3d_forward contract) never sets it.  Changing it invalidates the plan. */
nr_status nr_net_set_cfg_pair_identical(nr_net* h, int32_t enable);

/* ---- converted-weight exchange between handles (multi-GPU start-up) -----------------
 * The reference shards clips over processes and every process loads the checkpoints itself (scripts/neuroclips_video.py:
 * 94-138,238).  Here rank 0 loads + converts once (nr_net_load_tensor, nr_net_plan) and the converted bf16/fp32 device
 * buffers travel as ONE packed arena, device to device (RCCL broadcast over xGMI); the receivers never see fp32 host weights.
 *   nr_net_export_manifest  text description (state-dict key shapes + name/offset/bytes of every converted buffer); returns its
 *                           length (call with buf = NULL to size), arena_bytes receives the packed size
 *   nr_net_export_weights   packs the converted buffers into dst_dev (>= arena_bytes) on `stream`
 *   nr_net_import_weights   fresh handle (nothing loaded): adopts the arena; a following nr_net_plan with the SAME shape as the
 *                           exporter's finds every converted buffer and needs no host data */
int64_t nr_net_export_manifest(nr_net* h, char* buf, int64_t capacity, int64_t* arena_bytes);
nr_status nr_net_export_weights(nr_net* h, nr_stream stream, void* dst_dev, int64_t capacity);
nr_status nr_net_import_weights(nr_net* h, nr_stream stream, const char* manifest, int64_t manifest_bytes, const void* src_dev,
                                int64_t arena_bytes);

/* ---- debug / test hooks (activation taps by reference module name) --------------------------- */
nr_status nr_net_set_debug(nr_net* h, int32_t keep_all_activations);
int32_t nr_net_num_taps(const nr_net* h);
const char* nr_net_tap_name(const nr_net* h, int32_t i);
/* copies tap i (bf16 channels-last [rows][C]) to host as fp32; rows and C receive the shape */
nr_status nr_net_read_tap(nr_net* h, int32_t i, float* host_out, int64_t capacity, int32_t* rows, int32_t* C);

/* the launch plan of the current nr_net_plan(): one description string per enqueued op ("igemm ks=1 ... M= N= K=", "tattn_fused M= ...",
 * "ff_fused M= ...", "groupnorm ...", "" for boundary kernels); tests assert WHICH kernel serves a layer at a given shape */
int32_t nr_net_num_ops(const nr_net* h);
const char* nr_net_op_desc(const nr_net* h, int32_t i);

/* which GEMM kernel serves the big launches (gemm8p.hip, the 256-row ping-pong kernel): 0 never, 1 the shipped heuristic (default; env
 * NR_G8P), 2 whenever the shape is supported.  Process-wide; plan choices are read at nr_net_plan / launch time.  A/B tools and tests. */
void nr_g8p_set_mode(int32_t mode);
/* phases per k-tile of that kernel: 2 (default; tiles of <= 256 columns only) or 4.  A/B tools and tests (env NR_G8P_PHASES). */
void nr_g8p_set_phases(int32_t phases);
/* waves per 128-row workgroup of the fused FeedForward kernel (ffpanel.hip): 8 (default: 16-row waves, two per SIMD) or 4 (32-row waves,
 * one per SIMD; the round-3 form).  Results are bit-identical between the two.  A/B tools and tests (env NR_FF_WAVES). */
void nr_ff_set_waves(int32_t waves);

/* Leaf-module handles (kinds NR_KIND_LEAF_TRANSFORMER3D / NR_KIND_LEAF_TEMPORAL): Transformer3DModel.forward (attention.py:95-142) or
 * VanillaTemporalModule.forward (motion_module.py:79-86 -> TemporalTransformer3DModel.forward :134-158) on the reference's own tensors,
 * running exactly the launch sequence the engine uses for that module inside the U-Net at this shape.  Config: block_out_channels[0] = C,
 * num_levels 1, num_heads / cross_attention_dim / norm_num_groups (transformer) or motion_num_heads / motion_num_attention_blocks /
 * motion_pe_max_len (temporal).  State-dict keys: "m." + the module's own keys.  plan(batch, frames, h, w, ctx_len [0 for temporal]).
 *   x_dev fp32 [batch][C][F][h][w]; ctx_dev fp32 [batch][ctx_len][cross_attention_dim] (transformer only); out_dev like x_dev. */
nr_status nr_leaf_forward(nr_net* h, nr_stream stream, const float* x_dev, const float* ctx_dev, int32_t ctx_len, float* out_dev);

/* ---- single-op entry points (used by tests/ to check each kernel against the oracle) --------- */
/* The GEMM hooks below hand a launch on <= 512 rows with K a multiple of 640 to the panel-resident kernel (smallm.hip; the engine's rule for the
 * Linears of BasicTransformerBlock at the 4x4 level / the keyframe model, attention.py:256-300) after packing a fragment-major copy of w_dev on
 * the same stream, on every call.  NR_OP_FM_CACHE=1 keeps one copy per weight pointer instead (timing tools); nr_op_fm_cache_clear frees them. */
void nr_op_fm_cache_clear(void);
nr_status nr_op_gemm(nr_stream stream, const void* a_dev, int32_t lda, const void* w_dev, const float* bias_dev,
                     const void* res_dev, int32_t ldr, void* out_dev, int32_t ldo, int32_t M, int32_t N, int32_t K,
                     int32_t geglu);
/* out = [a0 | a1] . w^T (+bias) (+res): the two-source operand of the engine's conv(x, &skip, ...) / folded FeedForward as a 1x1 GEMM;
 * a0 [M][lda0] supplies channels [0, c0), a1 [M][lda1] channels [c0, c0 + c1); w bf16 [N][c0 + c1] */
nr_status nr_op_gemm2(nr_stream stream, const void* a0_dev, int32_t c0, int32_t lda0, const void* a1_dev, int32_t c1, int32_t lda1,
                      const void* w_dev, const float* bias_dev, const void* res_dev, int32_t ldr, void* out_dev, int32_t ldo, int32_t M,
                      int32_t N);
/* LayerNorm folded into the GEMM (engine: ln_linear): w_scaled[n][k] = gamma[k] W[n][k] (bf16), ln_c[n] = sum_k w_scaled[n][k],
 * bias_folded[n] = bias[n] + sum_k beta[k] W[n][k]; out = rstd_m (a . w_scaled^T - mean_m ln_c) + bias_folded with the row
 * statistics of `a` accumulated inside the kernel.  act: 0 none, 1 quick_gelu. */
nr_status nr_op_ln_gemm(nr_stream stream, const void* a_dev, int32_t lda, const void* w_scaled_dev, const float* ln_c_dev,
                        const float* bias_folded_dev, float eps, const void* res_dev, int32_t ldr, void* out_dev, int32_t ldo,
                        int32_t M, int32_t N, int32_t K, int32_t geglu, int32_t act);
/* every epilogue option of the Linear kernels in one call: folded LayerNorm (ln_c as above, or NULL), fp32 row-vector term
 * rowvec[((m / rowvec_div) % rowvec_mod) * rowvec_ld + n] (rowvec_mod 0 = no modulo; the time-embedding add of resnet.py:193-194 and
 * the temporal positional encoding pushed through to_q|k|v, motion_module.py:241-243,274-278), out = (acc + bias + rowvec) * out_scale
 * [quick_gelu] + res.  K = 320 on >= 4096 rows runs on the register-resident row-panel kernel (rowpanel.hip). */
nr_status nr_op_gemm_ex(nr_stream stream, const void* a_dev, int32_t lda, const void* w_dev, const float* bias_dev,
                        const float* ln_c_dev, float ln_eps, const float* rowvec_dev, int32_t rowvec_div, int32_t rowvec_mod,
                        int32_t rowvec_ld, const void* res_dev, int32_t ldr, void* out_dev, int32_t ldo, int32_t M, int32_t N,
                        int32_t K, int32_t geglu, int32_t act, float out_scale);
nr_status nr_op_conv3x3(nr_stream stream, const void* x0_dev, int32_t c0, const void* x1_dev, int32_t c1, int32_t nimg,
                        int32_t H, int32_t W, int32_t stride, int32_t ups, const void* w_dev, const float* bias_dev,
                        const float* rowvec_dev, int32_t rowvec_div, const void* res_dev, void* out_dev, int32_t Cout);
/* nr_op_conv3x3 for stride 1 / no upsample / one source, weight pre-arranged as [Cout][Cin/64][3][3][64] (the engine's layout for the
 * ResnetBlock convs: the K loop walks 64-channel chunks with the 9 taps innermost) */
nr_status nr_op_conv3x3_tap_inner(nr_stream stream, const void* x0, int32_t c0, int32_t nimg, int32_t H, int32_t W, const void* w,
                                  const float* bias, const float* rowvec, int32_t rowvec_div, const void* res, void* out, int32_t Cout);
nr_status nr_op_groupnorm(nr_stream stream, const void* x0_dev, int32_t c0, const void* x1_dev, int32_t c1, int32_t nimg,
                          int32_t hw, int32_t groups, const float* gamma_dev, const float* beta_dev, float eps,
                          int32_t silu, float* partial_ws_dev, void* out_dev);
nr_status nr_op_layernorm(nr_stream stream, const void* x_dev, void* out_dev, int32_t M, int32_t C,
                          const float* gamma_dev, const float* beta_dev, float eps, const float* pe_dev, int32_t pe_hw,
                          int32_t pe_F);
/* mode 0: spatial self ([nimg][L][3C] fused qkv), 1: cross (q [nimg][L][C], kv [nb_kv][Lk][2C], kv_div),
 * 2: temporal self (fused qkv [(b f)][hw][3C], sequence over f); mode | 8: e4m3 MFMA operands (modes 0 and 1, L >= 48) */
nr_status nr_op_attention(nr_stream stream, int32_t mode, const void* q_dev, const void* kv_dev, void* out_dev,
                          int32_t nimg, int32_t L, int32_t Lk, int32_t C, int32_t heads, int32_t frames, int32_t kv_div);

/* Fused FeedForward(GEGLU) + proj_out of the C = 320 level (ffpanel.hip; engine: feed_forward_proj_out):
 *   out = x + bc + [t | GEGLU(LayerNorm(t) W1^T + b1)] . Wc^T        (attention.py:129-140,297-299; motion_module.py:150-158,219-221)
 * with the weights in the engine's converted formats: w1 = value/gate-interleaved rows [8C][C] bf16 and b1 its bias in the same order,
 * gamma / beta the LayerNorm parameters, wc = [Wpo | Wpo Wff2] [C][5C] bf16, bc = bpo + Wpo bff2.  t, x, out: bf16 [M][C].
 * w1 == NULL re-uses the weight stream packed by the previous call. */
nr_status nr_op_ff_fused(nr_stream stream, const void* t_dev, const void* x_dev, void* out_dev, int32_t M, int32_t C,
                         const void* w1_geglu_dev, const float* gamma_dev, const float* beta_dev, const float* b1_geglu_dev,
                         const void* wc_dev, const float* bc_dev, float ln_eps);

/* One temporal-attention block of the C = 320 level in one launch (tattn.hip; engine: temporal_module):
 *   t <- t + to_out(softmax(q k^T / sqrt(d)) v),  [q | k | v] = (LayerNorm(t) + pe[frame]) [Wq | Wk | Wv]^T,  sequence = the 16 frames of a pixel
 * (motion_module.py:210-218,270-329; motion_module_new.py:201-287).  t: bf16 [nbatch * 16 * hw][320] ("(b f) (h w) c"), updated in place;
 * w*: bf16 [320][320] (nn.Linear weights); gamma fp32 [320]; gb fp32 [16][320] = LayerNorm bias + positional encoding of the frame;
 * bo fp32 [320] (to_out[0].bias).  wq == NULL re-uses the weight stream packed by the previous call. */
nr_status nr_op_tattn_fused(nr_stream stream, void* t_dev, int32_t nbatch, int32_t hw, const void* wq_dev, const void* wk_dev,
                            const void* wv_dev, const void* wo_dev, const float* gamma_dev, const float* gb_dev, const float* bo_dev,
                            float ln_eps);
/* One CROSS-attention block of the C = 320 level in one launch (xattn.hip, round 5; engine: spatial_transformer):
 *   t <- t + to_out(softmax(q K^T / sqrt(d)) V),  q = LayerNorm(t) Wq^T,  K | V = the context projections of the row's clip
 * (attention.py:281-290 norm2 -> attn2 -> + hidden_states, context repeated per frame :100; motion_module_new.py:201-287).
 * t: bf16 [nimg * hw][320] ("(b f) (h w) c"), updated in place, hw a multiple of 128; image i uses context i / img_per_ctx;
 * wq, wo: bf16 [320][320] (to_q.weight, to_out[0].weight); kv: bf16 [nctx * Lk][ldkv] with K in columns [0, 320) and V in [320, 640)
 * (the fused to_k | to_v projection of the context), Lk <= 80; gamma / beta fp32 [320] (norm2); bo fp32 [320] (to_out[0].bias).
 * wq == NULL re-uses the streams packed by the previous call. */
nr_status nr_op_xattn_fused(nr_stream stream, void* t_dev, int32_t nimg, int32_t hw, int32_t img_per_ctx, const void* wq_dev,
                            const void* wo_dev, const void* kv_dev, int32_t ldkv, int32_t Lk, int32_t nctx, const float* gamma_dev,
                            const float* beta_dev, const float* bo_dev, float ln_eps);
/* The same for sequences of `frames` = 16 or 32 frames (32: BASELINE config 5, motion_module temporal_position_encoding_max_len = 32):
 * t [nbatch * frames * hw][320], gb [frames][320], hw a multiple of 128 / frames. */
nr_status nr_op_tattn_fused_frames(nr_stream stream, void* t_dev, int32_t nbatch, int32_t frames, int32_t hw, const void* wq_dev,
                                   const void* wk_dev, const void* wv_dev, const void* wo_dev, const float* gamma_dev, const float* gb_dev,
                                   const float* bo_dev, float ln_eps);

/* The attention of one temporal-attention block ABOVE the C = 320 level up to (not including) to_out, one launch (tattnw.hip, round 6; engine:
 * temporal_module):  a[:, head] = softmax(q k^T / sqrt(d)) v per pixel over its 16 frames, [q | k | v] = (LayerNorm(t) + pe[frame]) [Wq | Wk | Wv]^T
 * (motion_module.py:210-218 norm -> VersatileAttention, :270-329; attention arithmetic motion_module_new.py:201-287), with the LayerNorm folded
 * as everywhere in this library: w_folded bf16 [3C][C] = gamma[k] W[n][k] (rows to_q | to_k | to_v), lnc fp32 [3C] = row sums of w_folded,
 * bias fp32 [3C] = sum_k beta[k] W[n][k], rowvec fp32 [16][3C] = pe[f] . W[n]^T.  t, a: bf16 [nbatch * 16 * hw][C] in "(b f) (h w) c" row order,
 * C = 640 (hw a multiple of 8) or 1280 (hw a multiple of 4), 8 heads.  w_folded == NULL re-uses the weight stream packed by the previous call at this C. */
nr_status nr_op_tattn_head(nr_stream stream, const void* t_dev, void* a_dev, int32_t nbatch, int32_t hw, int32_t C, const void* w_folded_dev,
                           const float* lnc_dev, const float* bias_dev, const float* rowvec_dev, float ln_eps);

/* The cross-attention of one BasicTransformerBlock ABOVE the C = 320 level up to (not including) to_out, one launch (xattnw.hip, round 6; engine:
 * spatial_transformer):  a = softmax(q K^T / sqrt(d)) V,  q = LayerNorm(t) Wq^T,  K | V = the context projections of the row's clip (attention.py:281-290,
 * context repeated per frame :100; motion_module_new.py:201-287), LayerNorm folded as everywhere in this library: wq_folded bf16 [C][C] = gamma[k] Wq[n][k],
 * lnc fp32 [C] = its row sums, bias fp32 [C] = sum_k beta[k] Wq[n][k].  t, a: bf16 [nimg * hw][C] in "(b f) (h w) c" row order, C = 640 or 1280 (8 heads), hw a
 * multiple of 64; image i uses context i / img_per_ctx; kv: bf16 [nctx * Lk][ldkv] with K in columns [0, C) and V in [C, 2C), Lk <= 80.
 * wq_folded == NULL re-uses the streams packed by the previous call at this C. */
nr_status nr_op_xattn_head(nr_stream stream, const void* t_dev, void* a_dev, int32_t nimg, int32_t hw, int32_t img_per_ctx, int32_t C,
                           const void* wq_folded_dev, const float* lnc_dev, const float* bias_dev, const void* kv_dev, int32_t ldkv, int32_t Lk,
                           int32_t nctx, float ln_eps);

#ifdef __cplusplus
}
#endif
#endif /* NEURONS_AMD_H */
