// IEEE-half TRAINING kernels: the training units (weight gradients, BatchNorm train forward / backward, row gather / scatter-add, the heads'
// small Linears) are compiled a second time with -DTL_F16_BUILD (treelearn_amd/build.py), like the conv units (tl_half.h): in that compilation
// "TL_BF16" stands for "the 16-bit type of this build" and every external symbol of the unit carries an _f16 suffix (the #defines below).
// The public entry points (default compilation) forward a TL_F16 call to their _f16 twin with the dtype code rewritten to TL_BF16.
// Reference: the training step runs under fp16 autocast + GradScaler (tools/training/train.py:32,40-44).
#pragma once
#include "tl_common.h"

#ifdef TL_F16_BUILD
#define tl_conv_wgrad_ws_floats tl_conv_wgrad_ws_floats_f16
#define tl_conv_wgrad tl_conv_wgrad_f16
#define tl_conv_wgrad_ref tl_conv_wgrad_ref_f16
#define tl_conv_wgrad_blk tl_conv_wgrad_blk_f16
#define tl_conv_wgrad_blk_ws_floats tl_conv_wgrad_blk_ws_floats_f16
#define tl_dev_wgrad_mode tl_dev_wgrad_mode_f16
#define g_wgrad_dma g_wgrad_dma_f16
#define g_wgrad_dense g_wgrad_dense_f16
#define g_wgrad_dense_min_rows g_wgrad_dense_min_rows_f16
#define g_wgrad_dense_gx g_wgrad_dense_gx_f16
#define g_wgrad_rows g_wgrad_rows_f16
#define tl_launch_wgrad_reduce tl_launch_wgrad_reduce_f16
#define tl_wgrad_dense_slots tl_wgrad_dense_slots_f16
#define tl_launch_wgrad_dense tl_launch_wgrad_dense_f16
#define tl_wgrad_rows_parts tl_wgrad_rows_parts_f16
#define tl_launch_wgrad_rows tl_launch_wgrad_rows_f16
#define tl_wgrad_in4_parts tl_wgrad_in4_parts_f16
#define tl_launch_wgrad_in4 tl_launch_wgrad_in4_f16
#define tl_wgrad_tinycout_parts tl_wgrad_tinycout_parts_f16
#define tl_launch_wgrad_tinycout tl_launch_wgrad_tinycout_f16
#define tl_launch_conv_tinycout tl_launch_conv_tinycout_f16
#define tl_linear_small_f32 tl_linear_small_f32_f16
#define tl_gather_rows tl_gather_rows_f16
#define tl_scatter_add_rows tl_scatter_add_rows_f16
#define tl_bn_train_finish tl_bn_train_finish_f16
#define tl_bn_train_bwd_from_parts tl_bn_train_bwd_from_parts_f16
#define tl_bn_ws_doubles tl_bn_ws_doubles_f16
#define tl_bn_train_stats tl_bn_train_stats_f16
#define tl_bn_train_bwd tl_bn_train_bwd_f16
#else
extern "C" {
int tl_conv_wgrad_f16(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin,
                      int Cout, float* gw, float* ws, tl_stream_t stream);
int tl_conv_wgrad_ref_f16(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* table, int64_t n_out, int64_t n_in, int K, int Cin,
                          int Cout, float* gw, float* ws, tl_stream_t stream);
int tl_conv_wgrad_blk_f16(const void* x, int64_t x_ld, const void* gout, int64_t g_ld, int dtype, const int32_t* blk_unit, const int32_t* blk_counter,
                          const int32_t* blk_halo, const uint32_t* blk_lrb, int64_t n, int Cin, int Cout, float* gw, int ref_layout, float* ws, tl_stream_t stream);
int tl_linear_small_f32_f16(const void* x, int64_t x_ld, int dtype, const void* w, int Cin, int Cout, int64_t n, float* out, int64_t out_ld, tl_stream_t stream);
int tl_gather_rows_f16(const void* in, int64_t in_ld, int dtype, int C, int64_t n_rows, const int64_t* idx, int64_t N, void* out, int64_t out_ld, tl_stream_t stream);
int tl_scatter_add_rows_f16(const void* g, int64_t g_ld, int dtype, int C, const int64_t* order, const int64_t* sorted_idx, int64_t N, int64_t n_rows, void* gin,
                            int64_t gin_ld, tl_stream_t stream);
int tl_bn_train_bwd_from_parts_f16(const void* x, int64_t ld, int x_dtype, const void* g, int64_t gld, int g_dtype, int64_t n, int C, const float* mean, const float* rstd,
                                   const float* scale, const float* shift, const double* part, int64_t nparts, float* dgamma, float* dbeta, void* dx, int64_t xld,
                                   const void* dx_add, int64_t dx_add_ld, tl_stream_t stream);
int tl_bn_train_stats_f16(const void* x, int64_t ld, int64_t n, int C, int dtype, const float* gamma, const float* beta, float eps, float momentum, double* ws,
                          float* mean, float* rstd, float* scale, float* shift, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          tl_stream_t stream);
int tl_bn_train_bwd_f16(const void* x, int64_t ld, int x_dtype, const void* dy, int64_t dld, int dy_dtype, int64_t n, int C, const float* mean, const float* rstd,
                        const float* scale, const float* shift, int relu, double* ws, float* dgamma, float* dbeta, void* dx, int64_t xld, const void* dx_add,
                        int64_t dx_add_ld, tl_stream_t stream);
}
// dtype code of a forwarded call: TL_F16 -> "the 16-bit type" of the _f16 compilation
static inline int tl_f16_code(int dtype) { return dtype == TL_F16 ? TL_BF16 : dtype; }
#endif
