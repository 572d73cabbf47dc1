// Shared by lamd_served.cpp (server) and lamd_client.cpp (client library): where the sections of a request lie in the shared block,
// and the two socket helpers (a struct over a stream socket, a file descriptor over SCM_RIGHTS).  include/lightning_amd_served.h is the protocol.
#pragma once
#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/types.h>
#include <unistd.h>

#include "../../include/lightning_amd_served.h"

namespace lamd_srv {

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// offsets of the request's input sections; returns the offset at which the output sections start (= total input bytes, aligned),
// or (size_t)-1 when the header is not sane
inline size_t layout(const lamd_srv_req &r, size_t off[LAMD_SRV_MAX_SECTIONS]) {
  if (r.n_sections > LAMD_SRV_MAX_SECTIONS) return (size_t)-1;
  size_t o = 0;
  for (uint32_t i = 0; i < r.n_sections; i++) {
    off[i] = o;
    if (r.section_len[i] > ((size_t)1 << 40)) return (size_t)-1;
    o = align16(o + (size_t)r.section_len[i]);
  }
  return o;
}

inline bool send_all(int fd, const void *p, size_t n) {
  const char *c = (const char *)p;
  while (n) {
    const ssize_t k = send(fd, c, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    c += k;
    n -= (size_t)k;
  }
  return true;
}
inline bool recv_all(int fd, void *p, size_t n) {
  char *c = (char *)p;
  while (n) {
    const ssize_t k = recv(fd, c, n, 0);
    if (k == 0) return false;
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    c += k;
    n -= (size_t)k;
  }
  return true;
}
// the request struct with one file descriptor attached
inline bool send_with_fd(int sock, const void *p, size_t n, int fd) {
  struct msghdr msg;
  memset(&msg, 0, sizeof msg);
  struct iovec iov = {(void *)p, n};
  char ctl[CMSG_SPACE(sizeof(int))];
  memset(ctl, 0, sizeof ctl);
  msg.msg_iov = &iov;
  msg.msg_iovlen = 1;
  msg.msg_control = ctl;
  msg.msg_controllen = sizeof ctl;
  struct cmsghdr *c = CMSG_FIRSTHDR(&msg);
  c->cmsg_level = SOL_SOCKET;
  c->cmsg_type = SCM_RIGHTS;
  c->cmsg_len = CMSG_LEN(sizeof(int));
  memcpy(CMSG_DATA(c), &fd, sizeof(int));
  for (;;) {
    const ssize_t k = sendmsg(sock, &msg, MSG_NOSIGNAL);
    if (k < 0 && errno == EINTR) continue;
    return k == (ssize_t)n;
  }
}
// receives exactly n bytes; *fd = the descriptor that came with them, or -1
inline bool recv_with_fd(int sock, void *p, size_t n, int *fd) {
  *fd = -1;
  char *dst = (char *)p;
  size_t got = 0;
  while (got < n) {
    struct msghdr msg;
    memset(&msg, 0, sizeof msg);
    struct iovec iov = {dst + got, n - got};
    char ctl[CMSG_SPACE(sizeof(int))];
    msg.msg_iov = &iov;
    msg.msg_iovlen = 1;
    msg.msg_control = ctl;
    msg.msg_controllen = sizeof ctl;
    const ssize_t k = recvmsg(sock, &msg, 0);
    if (k == 0) return false;
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    for (struct cmsghdr *c = CMSG_FIRSTHDR(&msg); c; c = CMSG_NXTHDR(&msg, c))
      if (c->cmsg_level == SOL_SOCKET && c->cmsg_type == SCM_RIGHTS) {
        if (*fd >= 0) close(*fd);
        memcpy(fd, CMSG_DATA(c), sizeof(int));
      }
    got += (size_t)k;
  }
  return true;
}

}  // namespace lamd_srv
