#include "pdmp3_oracle.h"
size_t orc_decode_buffer_like_cli(const unsigned char* mp3, size_t n, unsigned char* pcm, size_t pcm_cap, orc_tap* tap) { return 0; }
