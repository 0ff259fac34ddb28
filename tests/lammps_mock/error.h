// mock (see README.md): LAMMPS error.h -- error->all / error->one end the run; here they throw
#ifndef LMP_ERROR_H
#define LMP_ERROR_H
#include <stdexcept>
#include <string>
namespace LAMMPS_NS {
class LAMMPSException : public std::runtime_error { public: using std::runtime_error::runtime_error; };
class Error {
 public:
  [[noreturn]] void all(const std::string &file, int line, const std::string &str) { throw LAMMPSException("ERROR: " + str + " (" + file + ":" + std::to_string(line) + ")"); }
  [[noreturn]] void one(const std::string &file, int line, const std::string &str) { throw LAMMPSException("ERROR on proc 0: " + str + " (" + file + ":" + std::to_string(line) + ")"); }
  void warning(const std::string &, int, const std::string &) {}
};
}
#endif
