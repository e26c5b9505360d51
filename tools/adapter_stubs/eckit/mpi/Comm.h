// eckit::mpi::Comm (the members adapter/HaloExchangeMI355X.h and the reference's headers use; declarations only)
#pragma once
#include <cstddef>
#include <string>
#include <string_view>
#include <vector>
namespace eckit {
namespace mpi {
class Request {
public:
    Request();
};
class Status {
public:
    int source() const;
    int tag() const;
    int error() const;
};
class Comm {
public:
    virtual ~Comm();
    std::string name() const;
    std::size_t rank() const;
    std::size_t size() const;
    int communicator() const;
    void barrier() const;
    void abort(int errorcode = -1) const;
    Status wait(Request&) const;
    template <typename T>
    void broadcast(T& value, std::size_t root) const;
    template <typename Iter>
    void broadcast(Iter first, Iter last, std::size_t root) const;
    template <typename T>
    void allToAll(const std::vector<T>& send, std::vector<T>& recv) const;
    template <typename T>
    void allToAllv(const T* sendbuf, const int sendcounts[], const int sdispls[], T* recvbuf, const int recvcounts[],
                   const int rdispls[]) const;
    template <typename T>
    void allReduce(const T& send, T& recv, int op) const;
    template <typename T>
    void allReduceInPlace(T& sendrecv, int op) const;
    template <typename T>
    void allGather(const T& send, T* first, T* last) const;
    template <typename T>
    Request iReceive(T* recv, std::size_t count, int source, int tag) const;
    template <typename T>
    Request iSend(const T* send, std::size_t count, int dest, int tag) const;
    template <typename T>
    void send(const T* send, std::size_t count, int dest, int tag) const;
    template <typename T>
    Status receive(T* recv, std::size_t count, int source, int tag) const;
    Comm& split(int color, const std::string& name) const;
};
Comm& comm(const char* name = nullptr);
Comm& self();
void setCommDefault(const char* name);
bool hasComm(const char* name);
void deleteComm(const char* name);
void finaliseAllComms();
int sum();
int max();
int min();
}  // namespace mpi
}  // namespace eckit
