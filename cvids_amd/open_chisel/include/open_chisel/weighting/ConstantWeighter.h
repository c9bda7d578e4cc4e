#include "Weighter.h"
