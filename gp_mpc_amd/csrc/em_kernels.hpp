#pragma once
#include "mfma_f64.hpp"
namespace gpmpc {}
