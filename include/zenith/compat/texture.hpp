// texture.hpp — forwarding header of the MI355X drop-in: code written against the reference (#include "texture.hpp", /root/reference/texture.hpp)
// compiles against include/zenith/zenith.hpp when this directory is on the include path instead of the reference's sources.
#pragma once
#include "../zenith.hpp"
