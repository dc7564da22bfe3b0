#define VICTIM_NAME victim_pk
#define VICTIM_BFLY bfly_pk
#include "victim.inc"
