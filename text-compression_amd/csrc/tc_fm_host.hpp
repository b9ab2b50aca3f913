// tc_fm_host.hpp -- FM-index build / count / locate.
#pragma once
#include "tc_encode_host.hpp"
