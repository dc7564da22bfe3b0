// Internal interface between engine.hip (which owns the module handles) and dp_rccl.hip (which moves their weights between
// ranks): the device-resident storage of a handle's tensor table, in the table's (name-sorted, hence rank-independent) order.
#pragma once
#include <stddef.h>

#include <string>
#include <utility>
#include <vector>

namespace pf {

enum HandleKind { HANDLE_ENCODER = 0, HANDLE_PREDICTOR = 1, HANDLE_DECODER = 2, HANDLE_CTC = 3, HANDLE_VAD = 4 };

struct TensorSpan { std::string name; float* dev; size_t elems; bool set; };

// every tensor of the handle as it is stored in HBM (repacked / padded layouts included: all ranks build identical tables from
// identical configs, so the device images are interchangeable). Returns 0, or -1 for a null handle / unknown kind.
int handle_tensor_spans(int kind, void* handle, std::vector<TensorSpan>& out);
// the handle's weights were rewritten behind its back (a collective wrote into the spans): mark every tensor set and drop
// everything derived from the old values (resolved layer tables, cached operand planes, packed predictor weights)
int handle_weights_replaced(int kind, void* handle);

void set_error(const std::string& msg);

}  // namespace pf
