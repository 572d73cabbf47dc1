// Shared by lamd_served.cpp (server) and lamd_client.cpp (client library): where the sections of a request lie in the shared block,
// and the two socket helpers (a struct over a stream socket, a file descriptor over SCM_RIGHTS).  include/lightning_amd_served.h is the protocol.
#pragma once
#include <errno.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <string>

#include "../../include/lightning_amd_served.h"

namespace lamd_srv {

inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// Where the service listens when no --socket says otherwise: LAMD_SERVED_SOCKET, else $XDG_RUNTIME_DIR/lamd_served.sock (a directory only its
// user can write).  Never a world-writable directory: whoever binds the path first answers every commitment_signed and gossip check with "good".
inline bool default_socket(std::string *path, std::string *why) {
  const char *e = getenv("LAMD_SERVED_SOCKET");
  if (e && *e) { *path = e; return true; }
  const char *x = getenv("XDG_RUNTIME_DIR");
  if (x && *x == '/') { *path = std::string(x) + "/" LAMD_SRV_SOCKET_NAME; return true; }
  *why = "no socket path: set LAMD_SERVED_SOCKET (or XDG_RUNTIME_DIR) -- there is no default in a world-writable directory";
  return false;
}
// the process at the other end of a connected unix socket runs under `uid` (SO_PEERCRED: the kernel's word, taken at connect() / listen())
inline bool peer_uid_is(int fd, uid_t uid, uid_t *seen) {
  struct ucred cr;
  socklen_t len = sizeof cr;
  if (getsockopt(fd, SOL_SOCKET, SO_PEERCRED, &cr, &len) != 0 || len != sizeof cr) return false;
  if (seen) *seen = cr.uid;
  return cr.uid == uid;
}
// the uid a client expects its server under (and a server its clients): the caller's own, unless LAMD_SERVED_UID names another (a service account)
inline uid_t expected_peer_uid() {
  const char *e = getenv("LAMD_SERVED_UID");
  return e && *e ? (uid_t)strtoul(e, nullptr, 10) : geteuid();
}

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
