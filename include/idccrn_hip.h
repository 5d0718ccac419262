/* libidccrn_hip.so -- C ABI of the MI355X (gfx950) I-DCCRN-VAE enhancement hot path.
 *
 * The reference (iris1997jiatong/I-DCCRN-VAE) is pure PyTorch and has no FFI of its own; each entry
 * point below replaces the stock torch operator(s) behind one reference function (file:line cited
 * per entry, paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer (hipMalloc / torch.cuda memory) unless it says "host";
 *   - `stream` is a hipStream_t passed as void*; calls are asynchronous on it, never allocate device memory,
 *     never synchronise and are re-entrant across streams.  The ONLY process state the library keeps belongs to the
 *     cooperative recurrences (idv_lstm_rec_pers / _pers_f32 / _coop_f32 / idv_lstm_stack2_f32, idv_lstm_bptt_coop / _stack2): per device, the
 *     CU count, an event that orders cooperative launches of different streams, and a 256-byte host-mapped status word
 *     (see idv_coop_last_status);
 *   - return value: 0 ok, -1 invalid argument, -2 launch failure, -3 (IDV_ECOOP) an EARLIER cooperative recurrence ran into
 *     its spin bound (its outputs are NaN-poisoned) and that has not been acknowledged yet (idv_coop_last_status); nothing throws;
 *   - activations use the planar-J layout  act[ri][C][F][Jp]  (fp32):
 *       column j = b*Tp + tp, Tp = T+1, tp = t+1, tp==0 and tp>t_valid are zero guard columns,
 *       Jp >= B*Tp is the row stride; buffers need IDV_SLACK floats of slack in front and behind.
 *     A reference tensor x[B,C,F,T,2] is act.view(2,C,F,B,Tp)[..., 1:].permute(3,1,2,4,0).
 */
#ifndef IDCCRN_HIP_H
#define IDCCRN_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define IDV_ABI_VERSION 9
#define IDV_SLACK_FLOATS 256

int idv_abi_version(void);

/* Sticky status of the cooperative recurrences on the current device: -3 if one of them has run into its spin bound (a sibling
 * workgroup never became resident, e.g. on a CU-masked or shared GPU; its outputs were poisoned with NaN) since the status was
 * last cleared, else 0.  clear != 0 acknowledges and resets it.  The status is sticky like a device error: until it is
 * acknowledged EVERY cooperative entry of the device returns -3 without launching, so no unrelated caller can consume it.  The
 * operation that owns the time-out is the one whose stream was synchronised last: check after the synchronise (the Python
 * side does: ops.coop_check()).  The word is written by the device: synchronise the stream first for a definite answer. */
int idv_coop_last_status(int clear);
/* Workgroups a cooperative launch may use on the current device: multiProcessorCount minus 1/16 head room (240 on MI355X). */
int idv_coop_max_workgroups(void);

/* ---- weight preparation (device -> device, run once per parameter update) ------------------ */

/* K-chunk (in planar channels) the contraction kernel uses for `cin_used` complex input channels;
 * wfrag for a complex conv holds  roundup(2*Cout,128)/32 * roundup(2*cin_used, cck)*5 * 64 floats. */
int idv_cconv_cck(int cin_used);
/* Which cgemm_kernel instantiation idv_cconv2d_fwd launches for a layer shape: the template arguments
 * <MODE, WM, WN, MT_W, FO_T, JC_W, CCK> written as decimal digits (e.g. 1221324), -1 if unsupported; 1000001 = the
 * transposed conv with ONE output channel (last decoder block), which runs on the vector-ALU kernel of ctconv_c1_f32.hip.
 * Lets bench.py / profiles name the kernel a measured launch belongs to. */
int idv_cconv_config(int transposed, int cin_used, int Cout, int Fin);

/* ComplexBatchNormal statistics -> affine: model/complex_progress.py:168-209 (cbn).
 * moments: [5][C] = mean_r, mean_i, Vrr, Vri, Vii (Vrr/Vii already carry +1e-5, as the reference's
 * running buffers do).  fold[C][6] = Zrr,Zri,Zir,Zii, sr, si with  y = Z*x + s. */
int idv_cbn_fold(const float* moments, const float* gamma_rr, const float* gamma_ri, const float* gamma_ii,
                 const float* beta_r, const float* beta_i, int C, float* fold, void* stream);

/* Pack ComplexConv2d / causal_complex_conv2d weights (complex_progress.py:8-36; w_*: [Cout,Cin,5,2])
 * or (Causal)ComplexConvTranspose2d weights (:222-279; w_*: [Cin,Cout,5,2], transposed=1) into the
 * MFMA fragment order of the block matrix [[Wr,-Wi],[Wi,Wr]]; biases become (b_re-b_im, b_re+b_im)
 * (:16-18,:32-34).  fold (or NULL) is applied on the output side (eval-mode BN folded into the conv). */
int idv_pack_cconv(const float* w_re, const float* w_im, const float* b_re, const float* b_im, const float* fold,
                   int Cout, int Cin_total, int Cin_used, int transposed, float* wfrag, float* bias_out, void* stream);

/* Pack a row-major real matrix w[M][K] (+ bias[M] or NULL) for idv_pw_gemm;
 * wfrag holds roundup(M,128)/32 * roundup(K,8)/2 * 64 floats, bias_out roundup(M,128). */
int idv_pack_pw(const float* w, const float* bias, int M, int K, float* wfrag, float* bias_out, void* stream);

/* nn.LSTM input weights of ComplexLSTM (complex_progress.py:45-48): rows of the packed matrix are
 * [lstm_re gates | lstm_im gates], gate columns re-ordered for the recurrent kernel, bias = b_ih + b_hh.
 * w_ih_*: [4H][K].  Same sizes as idv_pack_pw with M = 8H. */
int idv_pack_lstm_ih(const float* w_ih_re, const float* b_ih_re, const float* b_hh_re, const float* w_ih_im,
                     const float* b_ih_im, const float* b_hh_im, int H, int K, float* wfrag, float* bias_out,
                     void* stream);
/* Recurrent weights w_hh_*: [4H][H] -> fragment order of the recurrent kernels: 2*4H*H floats of fp32 fragments
 * followed by the same number of bytes of split-bf16 (hi, lo) fragments; whh_frag holds 4*4H*H floats. */
int idv_pack_lstm_hh(const float* w_hh_re, const float* w_hh_im, int H, float* whh_frag, void* stream);

/* Windowed DFT / inverse-DFT matrices of torch.stft / torch.istft as used by STFT.forward /
 * ISTFT.forward (model/pvae_module.py:21-27, :38-42): hann(win) centred in n_fft, onesided.
 * w_fwd: [2F][win] row-major, w_inv: [win][2F] row-major, F = n_fft/2+1; env_inv: [n_fft + hop*(T-1)]
 * = 1 / overlap-added squared window (0 where the envelope is 0). */
int idv_make_dft(int n_fft, int win, int hop, int T, float* w_fwd, float* w_inv, float* env_inv, void* stream);

/* ---- forward operators ---------------------------------------------------------------------- */

/* ComplexConv2d.forward / causal_complex_conv2d.forward (complex_progress.py:16-22, :32-36) and
 * (causal_)ComplexConvTranspose2d.forward (:244-250, :275-279), kernel (5,2), stride (2,1), freq
 * padding 2; fused with torch.cat([p, skip], 1) of standard_DCCRN.forward (pvae_module.py:195;
 * x1 = skip, x1_div = num_samples for the repeated skips of pvae_module.py:2563-2567), with the
 * folded eval BatchNorm and PReLU (pvae_module.py:64-68, :88-93) when prelu_slope != NULL, and with
 * the train-mode moment sums (complex_progress.py:132-143) when stats != NULL (stats: [Cout][5]
 * doubles, zeroed by the caller: sum r, i, r*r, i*i, r*i over kept positions).  stats_work (or NULL) / stats_rep:
 * [stats_rep][Cout][5] zeroed doubles, stats_rep a power of two >= 2: the epilogue's atomic adds are spread over that many
 * replicas (chosen by workgroup) and folded into stats afterwards -- at 32 / 64 channels and ~10^4 workgroups the same-address
 * atomics otherwise cost more than the contraction (first encoder blocks, B = 32: 4.1 ms against 1.65).
 * tshift: -1 causal conv (taps x[t-1], x[t]) / any transposed conv of the model (taps x[t], x[t-1]); 0 non-causal conv
 * (taps x[t], x[t+1]); a transposed conv with tshift 0 reads (x[t+1], x[t]) -- with conjugate-transposed weights the
 * adjoint (data gradient) of the causal conv, as the non-causal conv is of the transposed conv (tests:
 * test_conv_adjoint_identity).  t_valid_out: frames kept. */
int idv_cconv2d_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div, const float* wfrag,
                    const float* bias, const float* prelu_slope, float* out, double* stats, double* stats_work, int stats_rep,
                    int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, void* stream);

/* idv_cconv2d_fwd with THREE real products per complex product (Gauss: t1 = Wr (xr + xi), t2 = (Wi - Wr) xr,
 * t3 = (Wr + Wi) xi; re = t1 - t3, im = t1 + t2 -- cgemm_gauss.hip), exact fp32 MFMA: the same reference lines
 * (model/complex_progress.py:16-22, :32-36, :244-250, :275-279), 25 % fewer multiplications than the reference's four real
 * convolutions.  idv_cconv_gauss_supported(C0, C1, Cout): >= 2 complex input channels, > 1 output channel, and with a second
 * source C0 even; callers use idv_cconv2d_fwd otherwise (the one-channel ends of the network).  idv_pack_cconv_gauss: weight
 * layouts as idv_pack_cconv; wfrag: idv_cconv_gauss_wfrag_floats(Cout, Cin_used) floats; epi: idv_cconv_gauss_epi_rows(Cout)
 * x 8 floats per output channel (Zrr, Zri, Zir, Zii, (Z b + s)_r, (Z b + s)_i, 0, 0) with b = (b_re - b_im, b_re + b_im) --
 * eval-mode ComplexBatchNormal (fold, complex_progress.py:161-209) is a real 2x2 map, so it cannot be folded into the three
 * weight planes and is applied by the epilogue when has_fold != 0 (then PReLU, pvae_module.py:58,82).  conj != 0 packs the
 * adjoint (data-gradient) operator: W_i negated, the caller passes the swapped channel roles as for idv_pack_cconv_adjoint.
 * addend (or NULL): planar [2][Cout][Fout][addend_Jp] holding B / addend_div utterances, added to the contraction
 * before bias / BN / PReLU: output utterance b takes addend utterance b / addend_div.  The convolution is linear in its input
 * channels, so for the repeated skips of pvae_module.py:2563-2567 the skip half is computed ONCE per utterance (a call with
 * x0 = skip, the skip rows of the weight, no bias) and added to each of its num_samples latent halves.
 * idv_cconv_gauss_config: the kernel instantiation as digits 3 MODE WM WN FO_T JC_W OCC (profiles; Cin = C0 + C1). */
int idv_cconv_gauss_supported(int C0, int C1, int Cout);
long long idv_cconv_gauss_wfrag_floats(int Cout, int cin_used);
int idv_cconv_gauss_epi_rows(int Cout);
int idv_cconv_gauss_config(int transposed, int Cin, int Cout, int Fin);
int idv_pack_cconv_gauss(const float* w_re, const float* w_im, const float* b_re, const float* b_im, const float* fold, int Cout,
                         int Cin_total, int Cin_used, int transposed, int conj, float* wfrag, float* epi, void* stream);
int idv_cconv2d_gauss_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div, const float* wfrag,
                          const float* epi, int has_fold, const float* prelu_slope, float* out, double* stats, double* stats_work,
                          int stats_rep, int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out,
                          const float* addend, int addend_div, int addend_Jp, void* stream);

/* Split-precision variant of idv_cconv2d_fwd (same reference lines): operands split into two bf16 (x = hi + lo),
 * w*x ~= w_hi*x_hi + w_hi*x_lo + w_lo*x_hi accumulated in fp32 on the bf16 MFMA (relative error ~2^-16 per
 * product; waveform parity vs the fp32 path ~1e-5, north_star tolerance 1e-3).  Needs C0 % 8 == 0, C1 % 8 == 0,
 * x1_div == 1, Cout >= 32, 16-byte aligned rows (idv_cconv_bf16_supported); callers fall back to
 * idv_cconv2d_fwd otherwise.  wfrag_bf16: idv_pack_cconv_bf16 (idv_cconv_bf16_wfrag_bytes bytes);
 * bias: the fp32 bias vector produced by idv_pack_cconv. */
long long idv_cconv_bf16_wfrag_bytes(int Cout, int cin_used);
int idv_cconv_bf16_supported(int transposed, int C0, int C1, int x1_div, int Cout);
int idv_cconv_bf16_config(int transposed, int Cout, int Fin);   /* <MODE, WM, WN, FO_T, JC_W, MT_W> as digits, for profiles */
int idv_pack_cconv_bf16(const float* w_re, const float* w_im, const float* fold, int Cout, int Cin_total, int Cin_used,
                        int transposed, void* wfrag, void* stream);
int idv_cconv2d_bf16x3_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, int x1_div, const void* wfrag_bf16,
                           const float* bias, const float* prelu_slope, float* out, double* stats, double* stats_work,
                           int stats_rep, int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out,
                           void* stream);   /* stats_work / stats_rep: as idv_cconv2d_fwd */

/* Split image: the inter-layer activation format of the bf16x3 path (eval mode).  bf16 elements,
 *   image[hi|lo][octet o][F][Jp][8],  element e of octet o = planar channel cc = 8*o + e = 2*ci + ri,
 * hi = value truncated to bf16, lo = round-to-nearest bf16 of the remainder (hi + lo reproduces the fp32 value to
 * ~2^-17, exactly the operand split idv_cconv2d_bf16x3_fwd applies while staging, so both entries give identical
 * results).  The lo plane starts lo_off elements (idv_cconv2d_img_fwd sources: 16-byte slots) after the hi plane;
 * callers keep >= 256 readable and writable bytes before each plane (every producer zeroes the 16-byte slot
 * 128 bytes before each plane; consumers read it for out-of-range frequency rows) and 4 KiB after them.
 * idv_cconv2d_img_fwd: same contraction, folded BN and PReLU epilogue as idv_cconv2d_bf16x3_fwd (reference
 * model/complex_progress.py:16-22, :244-250; pvae_module.py:64-68, :88-93) with image sources, writing planar fp32
 * (out_planar), an image (out_img) or both. */
int idv_cconv2d_fwd_img(const float* x0, int C0, const float* x1, int C1, int Jp1, const float* wfrag, const float* bias,
                        const float* prelu_slope, float* out_planar, void* out_img, long long out_lo_off_elems,
                        int transposed, int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out,
                        void* stream);   /* exact-fp32 idv_cconv2d_fwd (eval, x1_div 1) writing planar and/or image */
/* training forward from split images (bf16x3 training mode): planar fp32 y + the per-channel moments, as idv_cconv2d_fwd with
 * `stats`; sources as idv_cconv2d_img_fwd with src_is_image = 1 */
int idv_cconv2d_img_train_fwd(const void* x0_img, long long lo_off0, int C0, const void* x1_img, long long lo_off1, int C1,
                              const void* wfrag_bf16, const float* bias, float* out_planar, double* stats, double* stats_work,
                              int stats_rep, int transposed, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, void* stream);
int idv_cconv_img_config(int src_is_image, int transposed, int Cin, int Cout, int Fin);   /* template digits <MODE, WM, WN,
                        FO_T, JC_W, MT_W, IMGIN, AD> of the cgemm_bf16_kernel idv_cconv2d_img_fwd launches (profiles) */
int idv_planar_to_image(const float* x, int C, int F, int J, int Jp, void* img, long long lo_off_elems, void* stream);
/* the same with every utterance repeated rep times in a row (skip.repeat_interleave(num_samples, 0) of the two-phase decoder,
 * pvae_module.py:2563-2567, fused into the conversion): x has B utterances (pitch Jp_in), the image B * rep (pitch Jp) */
int idv_planar_to_image_repeat(const float* x, int C, int F, int B, int Tp, int Jp_in, int rep, void* img, long long lo_off_elems,
                               int Jp, void* stream);
int idv_image_to_planar(const void* img, long long lo_off_elems, int C, int F, int J, int Jp, float* x, void* stream);
int idv_cconv2d_img_fwd(int src_is_image /* 0: x0/x1 are planar fp32 (row stride Jp) */, const void* x0_img, long long lo_off0_slots, int C0, const void* x1_img, long long lo_off1_slots,
                        int C1, const void* wfrag_bf16, const float* bias, const float* prelu_slope, float* out_planar,
                        void* out_img, long long out_lo_off_elems, int transposed, int tshift,
                        int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, void* stream);

/* Last decoder block (Cout = 1): transposed conv re-associated so that the five frequency taps sit in the MFMA
 * M dimension (10 rows instead of 2), taps combined in the epilogue; split-bf16 arithmetic, eval mode only
 * (folded BN + PReLU; train-mode statistics use idv_cconv2d_fwd).  Same reference lines as idv_cconv2d_fwd.
 * Needs C0 % 8 == 0, C1 % 8 == 0, no repeated skips.  bias: bias_out of idv_pack_cconv (2 values used). */
long long idv_ctconv_c1_wfrag_bytes(int cin_used);
int idv_pack_ctconv_c1_bf16(const float* w_re, const float* w_im, const float* fold, int Cin_total, int Cin_used,
                            void* wfrag, void* stream);
int idv_ctconv_c1_bf16x3_fwd(const float* x0, int C0, const float* x1, int C1, int Jp1, const void* wfrag,
                             const float* bias, const float* prelu_slope, float* out, int Fin, int B, int Tp, int Jp,
                             int t_valid_out, void* stream);
int idv_ctconv_c1_img_fwd(const void* x0_img, long long lo_off0_slots, int C0, const void* x1_img, long long lo_off1_slots,
                          int C1, const void* wfrag, const float* bias, const float* prelu_slope, float* out, int Fin, int B,
                          int Tp, int Jp, int t_valid_out, void* stream);   /* same block, split-image sources */

/* out[m][j] = bias[m] + sum_k w[m][k] x[k][j] over K planes of stride Jp: ComplexDense.forward
 * (complex_progress.py:83-89, one call per real/imag linear), the LSTM input projections, and the
 * DFT / inverse DFT.  swap=1 writes out[((tp-1)*B + b)*ldo + m] instead of planar rows. */
int idv_pw_gemm(const float* x, int K, const float* wfrag, const float* bias, const float* prelu_slope, float* out,
                int M, int B, int Tp, int Jp, int t_valid, int swap, int ldo, void* stream);

/* Train-mode ComplexBatchNormal (complex_progress.py:131-160): sums -> moments[5][C] (mean_r, mean_i,
 * Vrr+eps, Vri, Vii+eps), running buffers updated in place (first call copies, later 0.9/0.1 blend),
 * fold[C][6] for idv_cbn_apply_prelu. count = B*F*T. */
int idv_cbn_finalize(const double* stats, double count, const float* gamma_rr, const float* gamma_ri,
                     const float* gamma_ii, const float* beta_r, const float* beta_i, int C, int first_call,
                     float momentum, float* running_mean_r, float* running_mean_i, float* Vrr, float* Vri,
                     float* Vii, float* moments, float* fold, void* stream);
/* Moment sums of a planar activation (stand-alone ComplexBatchNormal.forward, complex_progress.py:131-143);
 * stats [C][5] doubles, zeroed by the caller; same layout as the conv epilogue's. */
int idv_cbn_stats(const float* act, int C, int F, int B, int Tp, int Jp, int t_valid, double* stats, void* stream);
/* y = PReLU(Z*x + s) in place on a planar activation, guard columns stay zero. */
int idv_cbn_apply_prelu(float* act, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp, int Jp,
                        int t_valid, void* stream);

/* STFT.forward (pvae_module.py:21-27): x[B][L] -> frames[win][Jp] (reflect-padded, centred frames);
 * the DFT itself is idv_pw_gemm with the idv_make_dft matrix. */
int idv_stft_frames(const float* x, int B, int L, int n_fft, int win, int hop, int T, float* frames, int Tp, int Jp,
                    void* stream);
/* ISTFT.forward (pvae_module.py:38-42) tail: overlap-add of windowed inverse-DFT frames
 * frames[win][Jp] / envelope, trimmed to y[B][hop*(T-1)]. */
int idv_istft_ola(const float* frames, const float* env_inv, int B, int n_fft, int win, int hop, int T, int Tp,
                  int Jp, float* y, void* stream);

/* Mask branch of DCCRN_.forward (pvae_module.py:224-234) / decoder_twophase (:2594-2608):
 * predict = |X| tanh|M| exp(j(angle X + angle M)).  mask, X: planar [2][F][Jp] (X utterance b/x_div);
 * writes planar `pred` and the interleaved complex64 API tensor pred_c[B][F][T][2]. */
int idv_mask_apply(const float* mask, const float* X, int x_div, int JpX, float* pred, float* pred_c, int F, int B,
                   int T, int Tp, int Jp, void* stream);
/* Optional data normalisation of DCCRN_.forward (pvae_module.py:217-221 / :235-238; data_mean, data_std: [F][2] from
 * dataset/mean_*_spksplit.txt): out = (X - mean) / (std + 1e-6) with the imaginary part of bins 0 and F-1 zeroed, and the
 * inverse std * P + mean written planar (out) and interleaved [B][F][T][2] (out_c).  Eval path. */
int idv_datanorm(const float* X, const float* mean, const float* stdv, int F, int B, int T, int Tp, int Jp, float* out, void* stream);
int idv_datadenorm(const float* P, const float* mean, const float* stdv, int F, int B, int T, int Tp, int Jp, float* out,
                   float* out_c, void* stream);
/* gradients of the two (training with the reference's --data_norm): dX = dout / (std + 1e-6) with no gradient for the imaginary
 * parts of the first and last bin; dP = std * (dout + dout_c), either gradient may be NULL. */
int idv_datanorm_bwd(const float* dout, const float* stdv, int F, int B, int T, int Tp, int Jp, float* dX, void* stream);
int idv_datadenorm_bwd(const float* dout, const float* dout_c, const float* stdv, int F, int B, int T, int Tp, int Jp, float* dP,
                       void* stream);
/* Two-latent enhancement estimators of the evaluation script (i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:
 * real_and_imag_mask :85-101, complex_mask :104-116, phase_sensitive_mask :119-135, applied at :283-305): S / N = mean over the
 * ns sampled speech / noise spectra of an utterance, X = its noisy spectrum.  mode 0: per-part Wiener-like masks
 * (Sr^2 / (Sr^2 + Nr^2 + 1e-10)) Xr, same for the imaginary part; 1: S / (S + N + 1e-10) * X; 2: |S| / (|S| + |N| + 1e-10) *
 * cos(angle S - angle X) * |X| * exp(j angle S).  speech_c / noise_c: interleaved complex [B*ns][F][T][2] (the decoders'
 * `predict`); X: [B][F][T][2] with element strides (sb, sf, st, sr); out: planar [2][F][Jp] (feeds the ISTFT), out_c or NULL:
 * interleaved complex [B][F][T][2]. */
int idv_outtype_estimate(const float* speech_c, const float* noise_c, const float* X, long long sb, long long sf, long long st,
                         long long sr, int mode, int ns, int B, int F, int T, int Tp, int Jp, float* out, float* out_c,
                         void* stream);
/* planar [2][F][Jp] -> interleaved [B][F][T][2] (recon_type 'real_imag', pvae_module.py:245-253). */
int idv_planar_to_complex(const float* act, float* out_c, int F, int B, int T, int Tp, int Jp, void* stream);

/* ComplexLSTM.forward (complex_progress.py:50-74): four 2-layer LSTM passes, real = rr - ii,
 * imag = ir + ri.  x: planar [2][K][Jp]; out: planar [2][H][Jp].  flags bit 0: split-bf16 recurrence (H = 128).  wihN / bihN: idv_pack_lstm_ih of layer
 * N, whhN: idv_pack_lstm_hh of layer N.  work: idv_clstm_work_floats(H, B, T, Jp) floats. */
long long idv_clstm_work_floats(int H, int B, int T, int Jp);   /* 24*T*B*H + 4*B*H + scratch(H, B, Jp) + 4*H*Jp */
/* Persistent cooperative recurrence of one layer for H = 384 / 768 (the VAE encoders' 3*zdim / 6*zdim, reference
 * model/pvae_module.py:1819, :2160-2163), split-bf16 arithmetic: H/16 co-resident workgroups per weight set keep their
 * W_hh slice in registers for all T steps and exchange h_t through global memory (write-through stores, one arrive
 * counter per group of workgroups, no cache-wide fences; bounded spins; on a time-out the outputs are NaN).  idv_clstm_fwd
 * uses it when flags bit 0 is set and idv_lstm_pers_supported; flags bit 3 forces the per-step kernel.  g / g_run_z /
 * g_run_s / ldg address the gate pre-activations as G0 / G1 below, whh_frag: idv_pack_lstm_hh, work:
 * idv_lstm_pers_work_bytes(H, B) bytes, 16-byte aligned.  Outputs (at least one): hout [4 runs][T*B][H] fp32; kimg, the
 * K-major split image of h (slot ((run * H/8 + octet) * Jp + b*Tp + t + 1), lo plane kimg_lo_slots 16-byte slots behind
 * the hi plane) that idv_lstm_proj1_bf16x3 consumes. */
int idv_lstm_pers_supported(int H, int B);
long long idv_lstm_pers_work_bytes(int H, int B);
int idv_lstm_rec_pers(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout, int H,
                      int B, int T, void* work, void* kimg, long long kimg_lo_slots, int Tp, int Jp, float* gsave, float* csave,
                      void* stream);   /* gsave (== g) / csave: training forward, layouts of idv_clstm_fwd flags bit 2; or NULL */
/* Layer-1 input projection (nn.LSTM weight_ih_l1 of lstm_re / lstm_im, complex_progress.py:50-74) in split-bf16 from that
 * image: G1[run = 2z + s][(t, b)][4H].  wfrag_bf16: idv_pack_lstm_ih_bf16(w_ih_l1 re, im, H, K = H); bias: bih1 of
 * idv_pack_lstm_ih.  Needs 4H % 256 == 0 and H % 64 == 0. */
int idv_lstm_proj1_bf16x3(const void* himg, long long lo_off_slots, const void* wfrag_bf16, const float* bias, float* G1, int H,
                          int B, int T, int Tp, int Jp, void* stream);
/* The H = 128 recurrence (DCCRN-CL bottleneck) in exact fp32 on FOUR CUs per (run, 16-sequence tile) with the same hand-off
 * (lstm_coop_f32.hip): idv_clstm_fwd uses it in fp32 mode when idv_lstm_coop_f32_supported (H == 128, 16 * ceil(B/16) <= 240
 * workgroups; IDV_LSTM_COOP_F32=0 or flags bit 3 keep the one-CU register-resident kernel).  hout required; gsave (== g) /
 * csave as above; work: idv_lstm_coop_f32_work_bytes(H, B) bytes, 16-byte aligned. */
int idv_lstm_coop_f32_supported(int H, int B);
long long idv_lstm_coop_f32_work_bytes(int H, int B);
int idv_lstm_rec_coop_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout, int H,
                          int B, int T, void* work, float* gsave, float* csave, void* stream);
/* The same persistent cooperative recurrence in EXACT fp32 (lstm_pers_f32.hip; reference model/complex_progress.py:50-74 at the
 * hidden sizes of model/pvae_module.py:1819, :2160-2163): W_hh slices as fp32 in registers, v_mfma_f32_16x16x4_f32, h exchanged
 * as fp32 16-byte sc1 granules.  idv_clstm_fwd uses it in fp32 mode for H = 384 / 768 when idv_lstm_pers_f32_supported
 * (IDV_LSTM_PERS_F32=0 or flags bit 3 keep the per-step kernel).  Arguments as idv_lstm_rec_coop_f32. */
int idv_lstm_pers_f32_supported(int H, int B);
long long idv_lstm_pers_f32_work_bytes(int H, int B);
int idv_lstm_rec_pers_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh_frag, float* hout, int H,
                          int B, int T, void* work, float* gsave, float* csave, void* stream);
/* diagnostic (not part of the drop-in boundary): while a device buffer of 256 x 8 counters is registered, idv_lstm_rec_pers runs
 * an instrumented twin in which every workgroup accumulates core-clock cycles per phase (spin, -, barrier, loads + MFMA,
 * reduce + cell, drain + barrier, -, XCC id); NULL restores the production kernel.  tests/tools/lstm_phase_probe.py */
void idv_lstm_pers_set_profile(unsigned long long* prof_cycles);
/* flags bit 2 (training forward; with bit 0 only where idv_lstm_pers_supported, else EINVAL: exact-fp32 recurrence): the activated gates (i, f, g, o) and the cell states are kept
 * for idv_lstm_bptt.  work then holds idv_clstm_train_work_floats floats, laid out (TBH = T*B*H)
 *   [G0 16 TBH | G1 16 TBH | h0 4 TBH | h1 4 TBH | c0 4 TBH | c1 4 TBH | scratch]
 * G0: [z][T*B][8H] (z = real / imag input; columns [weight set s][4H]), G1: [run = 2z+s][T*B][4H], h/c: [run][T*B][H];
 * gate columns are ordered colp = ((u/16)*4 + gate)*16 + u%16. */
long long idv_clstm_train_work_floats(int H, int B, int T, int Jp);   /* 48*T*B*H + 4*B*H + scratch(H, B, Jp) + 4*H*Jp */
/* wih1_bf16 (may be NULL): idv_pack_lstm_ih_bf16 of layer 1; with it, flags bit 0 and the persistent recurrence, layer 0
 * hands h0 to the layer-1 projection as a split image (idv_lstm_proj1_bf16x3) instead of fp32 rows. */
int idv_clstm_fwd(const float* x, int K, const float* wih0, const float* bih0, const float* whh0, const float* wih1,
                  const float* bih1, const float* whh1, int H, int B, int T, int Tp, int Jp, float* work, float* out,
                  int flags, const void* wih1_bf16, void* stream);

/* Both layers of the H = 128 complex LSTM (reference ComplexLSTM.forward, model/complex_progress.py:50-74: nn.LSTM(num_layers = 2))
 * in ONE cooperative launch, exact fp32 (lstm_stack2_f32.hip): layer 1 runs one step behind layer 0 on its own four
 * CUs per (run, 16-sequence tile) and computes W_ih h0[t+1] in the hand-off latency of its own step t, so the hoisted layer-1
 * projection GEMM and the second recurrence launch disappear.  wih1_hh: idv_pack_lstm_hh applied to weight_ih_l1 (re, im) --
 * [4H][H] like W_hh; bias1: bih of idv_pack_lstm_ih for layer 1 ([2 sets][4H], gate-column order); h0, hout: [4 runs][T*B][H];
 * work: idv_lstm_stack2_f32_work_bytes bytes, 16-byte aligned; gsave1 ([run][T*B][4H]) / csave0 / csave1 ([4 runs][T*B][H]):
 * training forward (all three; layer 0's activated gates replace g in place -- the buffers idv_lstm_bptt reads) or all NULL.
 * Supported: H == 128 and 32 * ceil(B/16) <= idv_coop_max_workgroups() (IDV_LSTM_STACK2=0 turns it off).
 * idv_clstm_fwd2 = idv_clstm_fwd + wih1_hh (may be NULL): takes this path in fp32 mode (evaluation and flags bit 2) where
 * supported, idv_clstm_fwd's otherwise. */
int idv_lstm_stack2_f32_supported(int H, int B);
long long idv_lstm_stack2_f32_work_bytes(int H, int B);
int idv_lstm_stack2_f32(const float* g, long long g_run_z, long long g_run_s, int ldg, const float* whh0, const float* wih1_hh,
                        const float* whh1, const float* bias1, float* h0, float* hout, int H, int B, int T, void* work,
                        float* gsave1, float* csave0, float* csave1, void* stream);
int idv_clstm_fwd2(const float* x, int K, const float* wih0, const float* bih0, const float* whh0, const float* wih1,
                   const float* bih1, const float* whh1, const float* wih1_hh, int H, int B, int T, int Tp, int Jp, float* work,
                   float* out, int flags, const void* wih1_bf16, void* stream);

/* Split-precision (bf16x3) form of the layer-0 input projection inside idv_clstm_fwd (nn.LSTM weight_ih_l0 of
 * lstm_re / lstm_im applied to the real / imaginary input, complex_progress.py:50-74): ximg = K-major split image of
 * the 2*K input planes (idv_planar_to_kimage: octet o = planes 8o..8o+7, layout as "split image" with F = 1),
 * wfrag from idv_pack_lstm_ih_bf16 (idv_lstm_ih_bf16_bytes), bias = bih0 of idv_pack_lstm_ih, G = the first
 * 16*T*B*H floats of idv_clstm_fwd's work buffer; then call idv_clstm_fwd with flags bit 1 set.  Needs
 * 8H % 256 == 0 and K % 64 == 0 (idv_lstm_proj_bf16_supported). */
long long idv_lstm_ih_bf16_bytes(int H, int K);
int idv_lstm_proj_bf16_supported(int H, int K);
int idv_pack_lstm_ih_bf16(const float* w_ih_re, const float* w_ih_im, int H, int K, void* wfrag, void* stream);
int idv_planar_to_kimage(const float* x, int nvalid, int nplanes, int J, int Jp, void* img, long long lo_off_elems, void* stream);
/* bf16x3 form of idv_pw_gemm (swap = 0): out[m][Jp] = sum_k W[m][k] x[k][j] + bias[m], guard columns zero -- the
 * windowed DFT / inverse DFT of STFT.forward / ISTFT.forward (pvae_module.py:21-27, :38-42) and the two linears of
 * ComplexDense (complex_progress.py:83-89).  ximg: K-major split image with K rounded up to 64 planes
 * (idv_planar_to_kimage with nplanes = that, or idv_stft_frames_kimage = idv_stft_frames writing the image directly);
 * wfrag: idv_pack_pw_bf16 of the row-major [M][K] matrix (idv_pw_bf16_wfrag_bytes). */
/* the same contraction with the row-major transposed store out[(t*B + b)*ldo + m] over the valid (t, b): dh = W^T dG in the
 * form idv_lstm_bptt reads (bf16x3 training mode) */
int idv_pw_bf16x3_rows(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias, float* out_rows,
                       int M, int ldo, int B, int T, int Tp, int Jp, void* stream);
long long idv_pw_bf16_wfrag_bytes(int M, int K);
int idv_pack_pw_bf16(const float* w, int M, int K, void* wfrag, void* stream);
int idv_pw_bf16x3(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias, float* out, int M,
                  int B, int Tp, int Jp, int t_valid, void* stream);
int idv_stft_frames_kimage(const float* x, int B, int L, int n_fft, int win, int hop, int T, void* img, long long lo_off_elems,
                           int Tp, int Jp, void* stream);
int idv_lstm_proj_bf16x3(const void* ximg, long long lo_off_slots, int K, const void* wfrag_bf16, const float* bias, float* G,
                         int H, int B, int T, int Tp, int Jp, void* stream);

/* reparameterization (pvae_module.py:1832-1886) with the two randn draws supplied by the caller.
 * lat: planar LSTM output [2][Hl][Jp]; miu/log_sigma/delta are channel offsets off_miu/off_ls/off_dl
 * (zdim channels each).  eps_r/eps_i: [B][ns][T][zdim].  z: planar [2][zdim][Jpz], columns (b*ns+s)*Tp+tp. */
int idv_reparam(const float* lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim, const float* eps_r,
                const float* eps_i, int ns, int B, int T, int Tp, int Jp, float* z, int Jpz, void* stream);

/* ---- losses ---------------------------------------------------------------------------------- */

/* si_snr (model/sisnr_loss.py:7-19): three dot products per utterance; out[0] = -mean_b(snr).
 * src_div: source row = b / src_div (repeat over num_samples).  work: 3*B doubles (zeroed here). */
int idv_sisnr(const float* source, int src_ld, int src_div, const float* est, int est_ld, int B, int L, double* work,
              float* out, void* stream);
/* Enhancement inference.  idv_sisdr: compute_sisdr of utils/eval_metrics.py:49-64 on the device, one value per
 * utterance (out[B]); work: 3*B doubles.  idv_mean_over_samples: the mean over the num_samples sampled waveforms of one
 * utterance, i_dccrn_vae/nsvae_dccrn/test_se_cvaefinetune.py:309-311 (x: [B*ns][L] -> out [B][L]). */
int idv_sisdr(const float* ref, int ref_ld, const float* est, int est_ld, int B, int L, double* work, float* out, void* stream);
int idv_mean_over_samples(const float* x, int ns, int B, int L, float* out, void* stream);
/* multiple_recon_loss terms (model/nsvae_loss.py:775-797): out[0] = loss_cpx, out[1] = loss_mag
 * (the original magnitude uses the real part twice, nsvae_loss.py:783).  pred_c: interleaved
 * [B][F][T][2]; ori: element (b,f,t,ri) at ori[(b/ori_div)*sb + f*sf + t*st + ri*sr]. */
int idv_recon_loss(const float* pred_c, const float* ori, long long sb, long long sf, long long st, long long sr,
                   int ori_div, int B, int F, int T, double* work, float* out, void* stream);
/* Closed-form complex-Gaussian KL, mean over (b,t): cal_kl_arbi_prior
 * (model/pretrain_pvaes_loss.py:225-281, eps 1e-9) / cal_kl (model/nsvae_loss.py:275-328, eps 1e-10).
 * q1, q2: planar latents [2][Hn][Jpn] with channel offsets on_miu / on_ls / on_dl; q2 == NULL is the
 * standard prior (0, 0, 0).  work: 3 doubles.  out[0] = mean KL. */
int idv_ckl(const float* q1, int H1, int Jp1, int o1_miu, int o1_ls, int o1_dl, const float* q2, int H2, int Jp2,
            int o2_miu, int o2_ls, int o2_dl, int zdim, float eps, int B, int T, int Tp, double* work, float* out,
            void* stream);
/* One term of residual_loss (model/nsvae_loss.py:363-446: torch.mean((connct - connct2).pow(2)) per skip connection): the mean
 * over [B, C, F, T, 2] of (a[:, ca0:ca0+C] - b[:, cb0:cb0+C])^2 for two planar activations with Ca / Cb channels.  The reference
 * computes and RETURNS the term but never adds it to the loss it back-propagates (:460-466), so there is no gradient entry.
 * work: 3 doubles. */
int idv_msd(const float* a, int Ca, int ca0, int JpA, const float* b, int Cb, int cb0, int JpB, int C, int F, int B, int Tp,
            int t_valid, double* work, float* out, void* stream);
/* Minibatch mutual-information estimate of the CVAE / NVAE ELBO (complex_standard_vae_loss.mutual_information,
 * model/pretrain_pvaes_loss.py:129-159 on cal_gaussian_prob :64-127; the "- mi_weight * mi" term of cal_loss :334-343):
 * mean over (i, s, t) of log q(z[i,s,t] | x_i) - (logsumexp_j log q(z[i,s,t] | x_j) - log B).
 * lat: planar posterior [2][H][Jp] with (miu, log_sigma, delta) at channel offsets (column b*Tp + t + 1); z: the planar samples
 * [2][zdim][Jpz] of idv_reparam (column (b*ns + s)*Tp + t + 1).  work: idv_mi_work_floats floats, kept for idv_mi_bwd (it holds
 * d MI / d log q afterwards); acc: 1 double.  idv_mi_bwd: dz (same shape as z, written) and dlat (same shape as lat, += on the
 * three channel groups), each scaled by gout[0]; either may be NULL. */
long long idv_mi_work_floats(int B, int ns, int T, int zdim);
int idv_mi_fwd(const float* lat, int H, int Jp, int o_miu, int o_ls, int o_dl, const float* z, int Jpz, int zdim, int ns, int B,
               int T, int Tp, float eps, float* work, double* acc, float* out, void* stream);
int idv_mi_bwd(const float* lat, int H, int Jp, int o_miu, int o_ls, int o_dl, const float* z, int Jpz, int zdim, int ns, int B,
               int T, int Tp, float eps, const float* work, const float* gout, float* dlat, float* dz, void* stream);
/* miu_dis_loss term (model/nsvae_loss.py:349-360): sqrt(sum_{h,ri} mean_{b,t} (miu1 - miu2)^2). */
int idv_miu_dist(const float* q1, int H1, int Jp1, int off1, const float* q2, int H2, int Jp2, int off2, int zdim,
                 int B, int T, int Tp, double* work, float* out, void* stream);

/* ==== backward (gradient) entry points =========================================================================
 * The reference has no backward code of its own: torch.autograd differentiates the stock operators behind
 * `loss.backward()` in its train steps (supervised_dccrn/train.py:239-243, i_dccrn_vae/pretrained_vaes/train.py:296-301,
 * i_dccrn_vae/nsvae_dccrn/train_nsvae.py:557-561, train_second_phase_decoder.py:420-433).  Each entry below is the
 * gradient of the forward entry it names, for the same reference lines.  Planar gradients keep the layout invariant
 * (guard columns zero).  Data gradients of the conv blocks reuse idv_cconv2d_fwd with the adjoint weights. */

/* Adjoint weights for the DATA gradient of ComplexConv2d / ComplexConvTranspose2d (complex_progress.py:16-22, :244-250):
 * the parameter tensor read with the other layout (conv [Cout][Cin] as transposed-conv [Cin' = Cout][Cout' = Cin] and vice
 * versa), imaginary part negated, zero bias.  `transposed` / Cout / Cin_* describe the adjoint operator, which is then run by
 * idv_cconv2d_fwd: transposed = 1, tshift = 0 for the causal conv's gradient; transposed = 0, tshift = 0 and
 * t_valid_out = T for the causal transposed conv's. */
int idv_pack_cconv_adjoint(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used, int transposed,
                           float* wfrag, float* bias_out, void* stream);
/* the same operator as split-bf16 fragments for idv_cconv2d_bf16x3_fwd (training in bf16x3 mode; idv_cconv_bf16_wfrag_bytes) */
int idv_pack_cconv_bf16_adjoint(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used, int transposed,
                                void* wfrag, void* stream);

/* WEIGHT gradient of idv_cconv2d_fwd for one source of the channel concat (x: planar [2][Cx][Fin][Jp_x] = channels
 * ci_off .. ci_off+Cx of the weight tensor; dy: planar [2][Cout][Fout][Jp_dy], the gradient at the conv output, i.e. after
 * idv_cbn_bwd_apply in a train-mode block).  Writes dw_re / dw_im entries [*, ci_off:ci_off+Cx] (conv [Cout][Cin_total][5][2])
 * or [ci_off:ci_off+Cx, *] (transposed [Cin_total][Cout][5][2]).  Split-K over the B*Tp columns with deterministic
 * two-stage summation; work: idv_cconv_wgrad_work_floats(Cs, Cl, B, Tp) floats with (Cs, Cl) = (Cout, Cx) for a conv and
 * (Cx, Cout) for a transposed conv.  tshift as in idv_cconv2d_fwd. */
long long idv_cconv_wgrad_work_floats(int Cs, int Cl, int B, int Tp);
int idv_cconv2d_bwd_weight(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total, int transposed,
                           int tshift, int Fin, int B, int Tp, int Jp_x, int Jp_dy, float* work, long long work_floats,
                           float* dw_re, float* dw_im, void* stream);
/* the same contraction in split-bf16 arithmetic (both operands split on the fly, three v_mfma_f32_32x32x16_bf16 per product
 * block, fp32 accumulate): the weight gradient of the bf16x3 training mode; work: idv_cconv_wgrad_bf16_work_floats */
long long idv_cconv_wgrad_bf16_work_floats(int Cs, int Cl, int B, int Tp);
int idv_cconv2d_bwd_weight_bf16x3(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total, int transposed,
                                  int tshift, int Fin, int B, int Tp, int Jp_x, int Jp_dy, float* work, long long work_floats,
                                  float* dw_re, float* dw_im, void* stream);
/* idv_cconv2d_bwd_weight with THREE real contractions per complex channel pair (Gauss / Karatsuba: P1 = p u, P2 = q v,
 * P3 = (p - q)(u + v); dWr = P1 + P2, dWi = +-(P1 - P2 - P3) with (p, q) / (u, v) the real / imaginary planes of the two operands): the same
 * weight gradient (reference: torch.autograd of nn.Conv2d / nn.ConvTranspose2d, model/complex_progress.py:8-36, :222-279) with 25 %
 * fewer multiplications; arguments as idv_cconv2d_bwd_weight.  idv_cconv_wgrad_gauss_supported(Cs, Cl): Cs >= 128 and Cl >= 32
 * complex channels (Cs / Cl = the S / L side: conv (Cout, Cx), transposed conv (Cx, Cout); a narrower S side would leave the
 * kernel's 128-plane tile half empty); work: idv_cconv_wgrad_gauss_work_floats. */
int idv_cconv_wgrad_gauss_supported(int Cs, int Cl);
long long idv_cconv_wgrad_gauss_work_floats(int Cx, int Cout, int transposed, int Fin, int B, int Tp, int Jp_x, int Jp_dy);
int idv_cconv2d_bwd_weight_gauss(const float* x, int Cx, int ci_off, const float* dy, int Cout, int Cin_total, int transposed,
                                 int tshift, int Fin, int B, int Tp, int Jp_x, int Jp_dy, float* work, long long work_floats,
                                 float* dw_re, float* dw_im, void* stream);
/* bias gradients (b_re enters real as +, imag as +; b_im real as -, imag as +; complex_progress.py:16-18) from
 * idv_cbn_stats(dy): db_re = sum dy_r + sum dy_i, db_im = sum dy_i - sum dy_r. */
int idv_cconv2d_bwd_bias(const double* stats_dy, int Cout, float* db_re, float* db_im, void* stream);

/* Weight gradient of idv_pw_gemm (ComplexDense complex_progress.py:83-89; nn.LSTM weight_ih / weight_hh :45-48):
 * dw[rowmap(m)][k] (+)= sum_j dout[m][j] * x[k][j + shift]; dout / x planar rows of stride Jp_d / Jp_x; shift = -1 pairs
 * every column with its left neighbour (h_{t-1} of the recurrent weights; the guard column supplies h_{-1} = 0).
 * rowmap 1: rows arrive in the recurrent kernels' gate order (colp) and are written in torch's (gate*H + unit), M a
 * multiple of 4H.  work: idv_pw_wgrad_work_floats(M, K, J). */
long long idv_pw_wgrad_work_floats(int M, int K, int J);
int idv_pw_bwd_weight(const float* dout, int M, int Jp_d, const float* x, int K, int Jp_x, int J, int shift, float* work,
                      long long work_floats, float* dw, int ldw, int rowmap, int H, int accumulate, void* stream);
/* the same contraction in split-bf16 arithmetic (bf16x3 training mode: LSTM projection, dense and DFT weight gradients) */
int idv_pw_bwd_weight_bf16x3(const float* dout, int M, int Jp_d, const float* x, int K, int Jp_x, int J, int shift, float* work,
                             long long work_floats, float* dw, int ldw, int rowmap, int H, int accumulate, void* stream);
int idv_planar_rowsum(const float* x, int M, int Jp, int J, int accumulate, float* out, void* stream);   /* bias gradients */

/* Train-mode ComplexBatchNormal + PReLU (complex_progress.py:131-209, pvae_module.py:64-68, :88-93).
 * idv_cbn_apply_prelu_to: out-of-place idv_cbn_apply_prelu (training keeps the conv output y for the backward pass).
 * Backward, three steps: idv_cbn_bwd_reduce (per-channel sums [C][8] doubles of du = dz*PReLU'(u), du (x) y and the PReLU
 * slope term; data-parallel training all-reduces these), idv_cbn_bwd_finalize (moments = the [5][C] output of
 * idv_cbn_finalize, count = B*F*T of the whole batch; -> coef[C][12], d gamma_rr/ri/ii, d beta_r/i, dslope[1], each times
 * param_grad_scale: 1 normally, 1/world when the sums were all-reduced, because the ranks then all hold the gradient of
 * the SUM of their losses and the gradient all-reduce averages), idv_cbn_bwd_apply (dy = Z^T du + A (y - mu) + c). */
int idv_cbn_apply_prelu_to(const float* y, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp, int Jp,
                           int t_valid, float* out, void* stream);
int idv_cbn_bwd_reduce(const float* dz, const float* y, const float* fold, const float* prelu_slope, int C, int F, int B,
                       int Tp, int Jp, int t_valid, double* sums, void* stream);
int idv_cbn_bwd_finalize(const double* sums, double count, const float* moments, const float* gamma_rr, const float* gamma_ri,
                         const float* gamma_ii, int C, float* coef, float* dgamma_rr, float* dgamma_ri, float* dgamma_ii,
                         float* dbeta_r, float* dbeta_i, float* dslope, float param_grad_scale, void* stream);
int idv_cbn_bwd_apply(const float* dz, const float* y, const float* fold, const float* coef, const float* prelu_slope, int C,
                      int F, int B, int Tp, int Jp, int t_valid, float* dy, void* stream);
/* the out-of-place normalise + PReLU and the batch-norm backward apply, each also writing the split image of its result (bf16x3
 * training mode: the conv kernels read images; C % 4 == 0; image layout and edge slots as idv_planar_to_image) */
int idv_cbn_apply_prelu_to_img(const float* y, const float* fold, const float* prelu_slope, int C, int F, int B, int Tp, int Jp,
                               int t_valid, float* out, void* img, long long lo_off_elems, void* stream);
int idv_cbn_bwd_apply_img(const float* dz, const float* y, const float* fold, const float* coef, const float* prelu_slope, int C,
                          int F, int B, int Tp, int Jp, int t_valid, float* dy, void* img, long long lo_off_elems, void* stream);

/* idv_mask_apply backward (pvae_module.py:224-234): gradient w.r.t. the mask from the gradients of the planar prediction
 * and / or of the interleaved complex64 tensor (either may be NULL); dX (optional, x_div == 1 only): gradient w.r.t. the
 * input spectrum, needed when the waveform itself requires grad. */
int idv_mask_apply_bwd(const float* mask, const float* X, int x_div, int JpX, const float* dpred, const float* dpred_c, int F,
                       int B, int T, int Tp, int Jp, float* dmask, float* dX, void* stream);
/* Adjoint of repeating every utterance of a planar activation n times in a row (the skip connections of the fine-tuned
 * decoder, pvae_module.py:2563-2567): dx[row][b*Tp+tp] = sum_s drep[row][(b*n+s)*Tp+tp], rows = 2*C*F planar rows. */
int idv_repeat_batch_bwd(const float* drep, int n, int rows, int B, int Tp, int Jp_rep, int Jp, float* dx, void* stream);
/* interleaved [B][F][T][2] -> planar [2][F][Jp] (adjoint of idv_planar_to_complex; also packs a caller's complex tensor). */
int idv_complex_to_planar(const float* in_c, float* act, int F, int B, int T, int Tp, int Jp, void* stream);
/* idv_istft_ola backward (pvae_module.py:38-42): dy[B][hop*(T-1)] -> dframes[win][Jp]; the inverse-DFT adjoint is idv_pw_gemm
 * with the transposed matrix. */
int idv_istft_ola_bwd(const float* dy, const float* env_inv, int B, int n_fft, int win, int hop, int T, int Tp, int Jp,
                      float* dframes, void* stream);
/* idv_stft_frames backward (pvae_module.py:21-27, reflect padding included): dframes[win][Jp] -> dx[B][L]. */
int idv_stft_frames_bwd(const float* dframes, int B, int L, int n_fft, int win, int hop, int T, int Tp, int Jp, float* dx,
                        void* stream);
/* idv_reparam backward (pvae_module.py:1832-1886): dz planar [2][zdim][Jpz] -> += on the (miu, log_sigma, delta) channels of
 * dlat planar [2][Hl][Jp] (caller zeroes it). */
int idv_reparam_bwd(const float* lat, int Hl, int off_miu, int off_ls, int off_dl, int zdim, const float* eps_r,
                    const float* eps_i, int ns, int B, int T, int Tp, int Jp, const float* dz, int Jpz, float* dlat, void* stream);

/* Loss gradients.  grad_out / g_*: device scalars (the incoming gradient of the loss value).
 * idv_sisnr_bwd: work = the 3*B doubles idv_sisnr left behind; dest[B][L] = d loss / d estimate (sisnr_loss.py:7-19).
 * idv_recon_loss_bwd: d(loss_cpx, loss_mag) / d pred_c (nsvae_loss.py:775-797).
 * idv_ckl_bwd: d KL / d q1, += into dq1 (same layout as q1) (nsvae_loss.py:275-328, pretrain_pvaes_loss.py:225-281).
 * idv_miu_dist_bwd: loss_value = the forward result; += into dq1 and / or dq2 (nsvae_loss.py:349-360). */
int idv_sisnr_bwd(const float* source, int src_ld, int src_div, const float* est, int est_ld, int B, int L, const double* work,
                  const float* grad_out, float* dest, void* stream);
int idv_recon_loss_bwd(const float* pred_c, const float* ori, long long sb, long long sf, long long st, long long sr,
                       int ori_div, int B, int F, int T, const float* g_cpx, const float* g_mag, float* dpred_c, void* stream);
int idv_ckl_bwd(const float* q1, int H1, int Jp1, int o1_miu, int o1_ls, int o1_dl, const float* q2, int H2, int Jp2,
                int o2_miu, int o2_ls, int o2_dl, int zdim, float eps, int B, int T, int Tp, const float* grad_out, float* dq1,
                void* stream);
int idv_miu_dist_bwd(const float* q1, int H1, int Jp1, int off1, const float* q2, int H2, int Jp2, int off2, int zdim, int B,
                     int T, int Tp, const float* grad_out, const float* loss_value, float* dq1, float* dq2, void* stream);

/* ComplexLSTM backward (complex_progress.py:50-74), building blocks driven by the host mirror (ops.clstm_bwd):
 * idv_lstm_uncombine: planar output gradient -> per-run dh[4][T*B][H] (real = run0 - run3, imag = run2 + run1);
 * idv_pack_lstm_hh_bwd: W_hh [4H][H] x 2 -> fragments for the (16 x 4H) x (4H x H) step contraction, 2*4H*H floats;
 * idv_lstm_bptt: back-propagation through time of one layer, one launch per step; `gates` (activated i,f,g,o from the
 *   training forward, addressing as G0 / G1 above) is overwritten with the pre-activation gate gradients;
 *   work: idv_lstm_bptt_work_floats(H, B);
 * idv_rows_to_planar: src[(t*B + b)*ld + c0 + u] -> planar dst[u][b*Tp + t + 1] (guard columns zeroed);
 * idv_lstm_bias_grad: db[gate*H + unit] (+)= sum_j dGp[colp][j] for 4H planar rows. */
int idv_lstm_uncombine(const float* dout, int H, int B, int T, int Tp, int Jp, float* dh, void* stream);
int idv_pack_lstm_hh_bwd(const float* w_hh_re, const float* w_hh_im, int H, float* whhT, void* stream);
long long idv_lstm_bptt_work_floats(int H, int B);
int idv_lstm_bptt(float* gates, long long g_run_z, long long g_run_s, int ldg, const float* cstates, const float* dhout,
                  const float* whhT, int H, int B, int T, float* work, void* stream);
/* the same as ONE cooperative launch per layer for H = 128 (four workgroups per (run, 16-sequence tile), W_hh^T and the running
 * dc in registers, dA_t exchanged with the fence-free hand-off of idv_lstm_rec_pers); idv_lstm_bptt dispatches to it when
 * idv_lstm_bptt_coop_supported (IDV_LSTM_BPTT_COOP=0 keeps the per-step launches).  work: idv_lstm_bptt_coop_work_bytes. */
int idv_lstm_bptt_coop_supported(int H, int B);
/* BPTT of BOTH H = 128 layers in one cooperative launch (lstm_bptt_stack2_f32.hip; the backward twin of idv_lstm_stack2_f32):
 * layer 0 runs one step behind layer 1 and contracts dh0[t] = dA1[t] W_ih1 inside its recurrence, so the dh0 buffer and its
 * GEMM disappear.  g1: [run][T*B][4H] activated gates of layer 1 -> gate gradients; g0 (+ g0_run_z, g0_run_s, ldg0): layer 0's,
 * addressed as in idv_lstm_bptt; c1, c0: cell states [4][T*B][H]; dhout1: gradient arriving at layer 1's output; whhT1, wihT1,
 * whhT0: idv_pack_lstm_hh_bwd of weight_hh_l1, weight_ih_l1 ([4H][H] as well), weight_hh_l0; work:
 * idv_lstm_bptt_stack2_work_bytes bytes, 16-byte aligned.  Supported: H == 128, 32 * ceil(B/16) <= idv_coop_max_workgroups()
 * (IDV_LSTM_BPTT_STACK2=0 turns it off). */
int idv_lstm_bptt_stack2_supported(int H, int B);
long long idv_lstm_bptt_stack2_work_bytes(int H, int B);
int idv_lstm_bptt_stack2(float* g1, float* g0, long long g0_run_z, long long g0_run_s, int ldg0, const float* c1, const float* c0,
                         const float* dhout1, const float* whhT1, const float* wihT1, const float* whhT0, int H, int B, int T,
                         void* work, void* stream);
long long idv_lstm_bptt_coop_work_bytes(int H, int B);
int idv_lstm_bptt_coop(float* gates, long long g_run_z, long long g_run_s, int ldg, const float* cstates, const float* dhout,
                       const float* whhT, int H, int B, int T, void* work, void* stream);
int idv_rows_to_planar(const float* src, long long ld, int c0, int ncol, int B, int T, int Tp, int Jp, float* dst, void* stream);
int idv_lstm_bias_grad(const float* dGp, int H, int Jp, int J, int accumulate, float* db, void* stream);

/* ---- data-parallel train step: gradient bucket and optimiser as multi-tensor kernels (bucket.hip) --------------------------
 * The reference trains on one device with stock torch (`loss.backward(); optimizer.step()` with
 * torch.optim.Adam(lr, weight_decay=0.001): supervised_dccrn/train.py:109, 239-243; i_dccrn_vae/nsvae_dccrn/train_nsvae.py:200,
 * 557-561; i_dccrn_vae/nsvae_dccrn/train_second_phase_decoder.py:420-433; i_dccrn_vae/pretrained_vaes/train.py:296-301).
 * Bucket layout: a flat fp32 buffer in which tensor i occupies [off_i, off_i + numel_i), every off_i a multiple of 4 floats,
 * the gaps zero padding; table[n][3] (int64, device memory, sorted by off) = {device pointer of the tensor's first element
 * (0: absent), off_i, numel_i}; total = the padded length (multiple of 4); flat is 16-byte aligned.
 *   idv_bucket_gather : flat <- tensors (absent tensors and padding as zeros): the operand of the RCCL all-reduce;
 *   idv_bucket_scatter: tensors <- scale * flat (scale 1 after ReduceOp.AVG, 1 / world after a SUM);
 *   idv_bucket_adam   : one torch.optim.Adam step (no amsgrad / maximize) over all tensors of the table: parameters through
 *     ptable, gradients EITHER through gtable (same offsets) OR straight from a bucket gflat (e.g. the all-reduced one, so the
 *     scatter pass disappears), multiplied by grad_scale; exp_avg / exp_avg_sq in the bucket layout;
 *     bias_correction1 = 1 - beta1^t, bias_correction2_sqrt = sqrt(1 - beta2^t) for the step count t of this step; the betas are
 *     doubles because torch forms 1 - beta in double before rounding (1 - 0.999f in float is 1.3e-5 off).  A table row with a
 *     NULL parameter pointer is skipped (a parameter that received no gradient keeps its value and moments, as in torch). */
int idv_bucket_gather(const long long* table, int n, long long total, float* flat, void* stream);
int idv_bucket_scatter(const long long* table, int n, long long total, const float* flat, float scale, void* stream);
int idv_bucket_adam(const long long* ptable, const long long* gtable, const float* gflat, float* exp_avg, float* exp_avg_sq,
                    int n, long long total, float lr, double beta1, double beta2, float eps, float weight_decay,
                    float bias_correction1, float bias_correction2_sqrt, float grad_scale, void* stream);

/* ---- complex conv / transposed conv with Winograd-transformed frequency taps (cgemm_wino.hip) -------------------------------
 * The encoder / decoder blocks' contractions (model/complex_progress.py:8-36, :222-279) with 7 instead of 10 real products per
 * input channel and pair of rows (F(2,3) on the even, F(2,2) on the odd frequency taps) on top of the three-product complex form
 * of idv_cconv2d_gauss_fwd; same result up to the rounding of the transforms.  idv_pack_cconv_wino: weights as
 * idv_pack_cconv_gauss (transposed / conj conventions), wfrag of idv_cconv_wino_wfrag_floats floats; the epilogue table (epi,
 * has_fold) is idv_pack_cconv_gauss's.  idv_cconv2d_wino_fwd: arguments as idv_cconv2d_gauss_fwd with x1_div = 1 and both
 * sources at pitch Jp (Jp % 4 == 0, 16-byte aligned).  IDV_WINO=0 turns it off (IDV_WINO_CONV=0: the conv form only). */
int idv_cconv_wino_supported(int transposed, int C0, int C1, int Cout, int Fin);
long long idv_cconv_wino_wfrag_floats(int transposed, int Cout, int cin_used);
int idv_cconv_wino_config(int transposed, int Cin, int Cout);
int idv_pack_cconv_wino(const float* w_re, const float* w_im, int Cout, int Cin_total, int Cin_used, int transposed, int conj,
                        float* wfrag, void* stream);
int idv_cconv2d_wino_fwd(const float* x0, int C0, const float* x1, int C1, const float* wfrag, const float* epi, int has_fold,
                         const float* prelu_slope, float* out, double* stats, double* stats_work, int stats_rep, int transposed,
                         int tshift, int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, const float* addend,
                         int addend_div, int addend_Jp, void* stream);

/* ---- transposed conv with Winograd-transformed frequency AND time taps (cgemm_tw.hip) ----------------------------------------
 * The decoder block's contraction (model/complex_progress.py:222-279) as idv_cconv2d_wino_fwd(transposed = 1), with the two time
 * taps in F(2,2) form over pairs of output columns: 3 instead of 4 real products per column pair and channel pair, i.e. 3/4 x 7/10
 * x 3/4 = 0.39 of the reference's real products.  Same result up to the rounding of the transforms.  idv_pack_cconv_tw re-orders
 * the fragments of idv_pack_cconv_wino(transposed = 1) (wino_frag) into idv_cconv_tw_wfrag_floats floats; epi / has_fold from
 * idv_pack_cconv_gauss; statistics and addend as idv_cconv2d_wino_fwd.  Sources at pitch Jp (Jp % 4 == 0, 16-byte aligned),
 * C0 % 8 == 0 with a second source.  IDV_TW=0 (Python side) keeps idv_cconv2d_wino_fwd. */
int idv_cconv_tw_supported(int C0, int C1, int Cout, int Fin);
long long idv_cconv_tw_wfrag_floats(int Cout, int cin_used);
int idv_pack_cconv_tw(const float* wino_frag, int Cout, int cin_used, float* tw_frag, void* stream);
int idv_ctconv2d_tw_fwd(const float* x0, int C0, const float* x1, int C1, const float* wfrag, const float* epi, int has_fold,
                        const float* prelu_slope, float* out, double* stats, double* stats_work, int stats_rep, int tshift,
                        int Cout, int Fin, int B, int Tp, int Jp, int t_valid_out, const float* addend, int addend_div,
                        int addend_Jp, void* stream);

/* ---- conv with Winograd-transformed frequency AND time taps (cgemm_tw2.hip) ---------------------------------------------------
 * The encoder block's contraction (model/complex_progress.py:8-36) as idv_cconv2d_wino_fwd(transposed = 0) with the two time taps
 * in F(2,2) form (0.39 of the reference's real products); one source, no addend.  idv_pack_cconv_tw2 re-orders the fragments of
 * idv_pack_cconv_wino(transposed = 0) into idv_cconv_tw2_wfrag_floats floats; epi / has_fold from idv_pack_cconv_gauss;
 * statistics as idv_cconv2d_wino_fwd.  Source at pitch Jp (Jp % 4 == 0, 16-byte aligned). */
int idv_cconv_tw2_supported(int Cin, int Cout, int Fin);
long long idv_cconv_tw2_wfrag_floats(int Cout, int cin_used);
int idv_pack_cconv_tw2(const float* wino_frag, int Cout, int cin_used, float* tw_frag, void* stream);
int idv_cconv2d_tw_fwd(const float* x0, int Cin, const float* wfrag, const float* epi, int has_fold, const float* prelu_slope,
                       float* out, double* stats, double* stats_work, int stats_rep, int tshift, int Cout, int Fin, int B, int Tp,
                       int Jp, int t_valid_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
