// gph_comm.h -- the cross-rank exchange of the engine (RCCL all-gather on the engine's stream, or a host
// shared-memory exchange for ranks that share a GPU): declared in the public header, implemented in gph_comm.cpp.
#pragma once
#include "../../include/gphocs_hip.h"
