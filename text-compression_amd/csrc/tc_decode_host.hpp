// tc_decode_host.hpp -- decode path (inverse RLE / MTF / BWT).
#pragma once
#include "tc_encode_host.hpp"
