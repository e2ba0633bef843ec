// Parameter structs are defined once, in the public C header.
#pragma once
#include "../../include/ief_hip.h"
