// Front-end check of adapter/HaloExchangeMI355X.h (header-only, all templates): instantiates what parallel::HaloExchange's callers
// instantiate -- execute / execute_adjoint for the four POD types, ranks 1 - 3, both parallel dimensions (Packer.cc:71-95,
// HaloExchange.h:151-290), the four setup overloads (HaloExchange.h:47-58).  Never linked, never run.
#include "HaloExchangeMI355X.h"

namespace {
template <typename T>
void instantiate(atlas::parallel::HaloExchangeMI355X& hx, atlas::array::Array& a) {
    hx.execute<T, 1>(a);
    hx.execute<T, 2>(a, true);
    hx.execute<T, 3>(a, false);
    hx.execute<T, 2, atlas::array::LastDim>(a);
    hx.execute_adjoint<T, 1>(a);
    hx.execute_adjoint<T, 2>(a, true);
    hx.execute_adjoint<T, 3, atlas::array::LastDim>(a);
}
}  // namespace

void atlas_amd_check_halo_exchange(atlas::array::Array& a, const int part[], const atlas::idx_t ridx[]) {
    atlas::parallel::HaloExchangeMI355X hx("check");
    hx.setup(part, ridx, 0, 10);
    hx.setup("world", part, ridx, 0, 10);
    hx.setup(part, ridx, 0, 10, 5);
    hx.setup("world", part, ridx, 0, 10, 5);
    instantiate<int>(hx, a);
    instantiate<long>(hx, a);
    instantiate<float>(hx, a);
    instantiate<double>(hx, a);
    (void)hx.name();
}
