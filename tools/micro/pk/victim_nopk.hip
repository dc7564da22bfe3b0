#define VICTIM_NAME victim_nopk
#define VICTIM_BFLY bfly_nopk
#include "victim.inc"
