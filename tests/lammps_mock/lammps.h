#include "pointers.h"
