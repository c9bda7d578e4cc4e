#include "Chisel.h"
