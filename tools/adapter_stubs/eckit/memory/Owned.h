// eckit::Owned as the reference's headers use it (atlas/util/Object.h:18): an intrusive reference count.  Declarations only.
#pragma once
#include <cstddef>
namespace eckit {
class Owned {
public:
    Owned();
    Owned(const Owned&)            = delete;
    Owned& operator=(const Owned&) = delete;
    virtual ~Owned();
    void attach() const;
    void detach() const;
    std::size_t owners() const;
    void lock() const;
    void unlock() const;
};
}  // namespace eckit
