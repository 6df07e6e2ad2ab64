// stand-in: see se3.hpp
#pragma once
#include "se3.hpp"
#include "sim3.hpp"
