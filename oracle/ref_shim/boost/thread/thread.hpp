// stand-in: see boost/thread.hpp
#pragma once
#include "boost/thread.hpp"
