// declarations of tools/experiments/lintile.hip (not part of the product library)
#pragma once
int lin_tile_launch(const LinChainParams& c, hipStream_t st);
size_t lin_tile_stream_bytes(int layers);
int lin_tile_pack_launch(const float* w, int layer, float scale, int layers, bf16_t* stream, hipStream_t st);
