// fdr_image_io.hpp -- the image codecs and colour conversions moved to include/fdr_cv.hpp (namespace fdr_io), where the
// cv:: free functions of the drop-in surface use them too; this header keeps the CLIs' include line working.
#pragma once
#include "fdr_cv.hpp"
