#pragma once
#include "kernels_modexp.hpp"
