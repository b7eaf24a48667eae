// TEST INFRASTRUCTURE: host storage for the kernels' dynamic LDS (`extern __shared__ T name[]` in the HIP sources becomes
// `extern T name[]` in the copies the host build compiles; 160 KB = the CDNA4 LDS size).
namespace qtts {
alignas(16) float sm_ad[160 * 1024 / 4];                 // attention.hip   (attn_decode)
alignas(16) unsigned char smem_sk[160 * 1024];           // skinny.hip
alignas(16) unsigned char smem_gw[160 * 1024];           // gemm_tap.hip    (wide-K variant)
alignas(16) unsigned char smem_t2[160 * 1024];           // gemm_tap.hip    (tap-reuse kernel, round 2)
alignas(16) unsigned char smem_gd[160 * 1024];           // gemm_tap.hip    (LDS-DMA kernel, round 4)
alignas(16) unsigned char smem_gr[160 * 1024];           // gemm_tap.hip    (ring kernel, round 6)
alignas(16) float sm_fc[160 * 1024 / 4];                 // elementwise.hip (final conv)
alignas(16) unsigned char smem_ru[160 * 1024];           // resunit.hip     (fused residual unit, round 3)
alignas(16) unsigned char smem_cl[160 * 1024];           // cp_layer.hip    (one launch per code-predictor layer, round 6)
}  // namespace qtts
