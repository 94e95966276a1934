// declarations of tools/experiments/xffn.hip (not part of the product library)
#pragma once
size_t xffn_stream_bytes();
size_t xffn_bias_bytes();
int xffn_pack_launch(const float* w, int which, bf16_t* stream, hipStream_t st);      // which as ffn_pack_launch
int xffn_pack_bias_launch(const float* b1, float* out, hipStream_t st);
int xffn_launch(const FfnParams& f, hipStream_t st);
